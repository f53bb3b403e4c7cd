"""
-m "not gpu": the C-ABI library loads and exports every symbol include/grx.h declares, the ctypes
table mirrors the header, argument validation works without touching a device, and the product
path fails loudly (no CPU fallback) when there is no GPU.  No compute calls here.
"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    header = open(os.path.join(ROOT, 'include', 'grx.h')).read()
    return sorted(set(re.findall(r'\b(grx_[a-z0-9_]+)\s*\(', header)))


def test_library_is_built_in_tree():
    from graphrole_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run __graft_entry__.build() first'
    assert os.path.dirname(_lib.LIB_PATH) == os.path.join(ROOT, 'graphrole_amd')


def test_every_header_symbol_is_exported_and_bound():
    from graphrole_amd import _lib
    lib = _lib.load()
    declared = _declared()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in grx.h but not exported by libgrx.so'
        assert name in _lib.EXPORTED_SYMBOLS, f'{name} has no ctypes signature in graphrole_amd/_lib.py'
    for name in _lib.EXPORTED_SYMBOLS:
        assert name in declared, f'{name} bound in _lib.py but missing from include/grx.h'


def test_version_and_error_string():
    from graphrole_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "grx.h")).read()
    assert lib.grx_version() == int(re.search(r"#define\s+GRX_VERSION\s+(\d+)", header).group(1))
    assert isinstance(lib.grx_last_error(), bytes)
    assert lib.grx_profile_kernel_count() >= 20
    names = {lib.grx_profile_kernel_name(i).decode() for i in range(lib.grx_profile_kernel_count())}
    assert {'aggregate_kernel', 'nmf_w_pass_kernel', 'scatter_kernel', 'triangle_count_kernel'} <= names


def test_argument_validation_needs_no_device():
    """Bad arguments are rejected before any HIP call (status GRX_ERR_INVALID = -1 -> ValueError)."""
    from graphrole_amd import _lib
    lib = _lib.load()
    # frac outside (0,1): the reference raises ValueError('must specify frac in interval (0, 1)')
    rc = lib.grx_vertical_log_bin(10, 1, None, 10, ctypes.c_double(1.5), None, 10, None, None, 0, None)
    assert rc == -1
    assert b'frac' in lib.grx_last_error()
    with pytest.raises(ValueError, match='frac'):
        _lib.call('grx_vertical_log_bin', 10, 1, None, 10, ctypes.c_double(0.0), None, 10, None, None, 0, None)
    rc = lib.grx_aggregate(None, None, None, 3, None, 4, 0, 10, None, None, 10, None)             # no plan
    assert rc == -1 and b'plan' in lib.grx_last_error()
    # a plan for a graph without long rows is pure host work
    import numpy as np
    row_ptr = np.arange(0, 33, 3, dtype=np.int64)                                                 # 10 rows x 3
    plan = ctypes.c_void_p()
    assert lib.grx_aggregate_plan_create(10, row_ptr.ctypes.data_as(ctypes.c_void_p), ctypes.byref(plan)) == 0
    n_long, n_blocks, lanes = ctypes.c_int64(-1), ctypes.c_int64(-1), ctypes.c_int(-1)
    assert lib.grx_aggregate_plan_info(plan, ctypes.byref(n_long), ctypes.byref(n_blocks), ctypes.byref(lanes)) == 0
    assert (n_long.value, n_blocks.value, lanes.value) == (0, 0, 4)
    rc = lib.grx_aggregate(plan, None, None, 3, None, 3, 0, 10, None, None, 10, None)             # odd ldr
    assert rc == -1 and b'ldr' in lib.grx_last_error()
    rc = lib.grx_aggregate_minmax(plan, None, None, 3, None, 4, 0, 11, None, None, 10, None)      # bad row range
    assert rc == -1 and b'row range' in lib.grx_last_error()
    assert lib.grx_aggregate_plan_set_lanes(plan, 5) == -1
    lib.grx_aggregate_plan_destroy(plan)
    rc = lib.grx_row_sums(5, None, None, None, 0, 3, 9, None, None)                               # bad row range
    assert rc == -1
    assert lib.grx_gram(10, 500, None, 10, 0, 10, None, 500, None, None, 0, None) == -4           # unsupported F
    # workspace queries are pure host arithmetic
    assert lib.grx_log_bin_workspace_bytes(1_000_000, 12) > 2 * 12 * 8_000_000
    assert lib.grx_sort_workspace_bytes(1_000_000, 12) > 12 * 8_000_000
    assert lib.grx_nmf_workspace_bytes(1_000_000, 20, 6) > 0


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    import networkx as nx
    from graphrole_amd import RecursiveFeatureExtractor, backend
    from graphrole_amd._lib import GrxError
    backend.use(None)
    with pytest.raises(GrxError, match='no CPU fallback'):
        RecursiveFeatureExtractor(nx.path_graph(4)).extract_features()
    from graphrole_amd.roles import factor
    import numpy as np
    with pytest.raises(GrxError, match='no CPU fallback'):
        factor.get_nmf_decomposition(np.random.rand(30, 4), 2)


def test_product_code_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under graphrole_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'graphrole_amd')):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, re.M), f'{fn} imports oracle'
                assert 'liboracle' not in text, f'{fn} references liboracle'


def test_every_binding_call_passes_the_declared_number_of_arguments():
    """Static guard (no GPU needed): each `_lib.call('grx_x', ...)` / `lib.grx_x(...)` in the host package passes as
    many arguments as the ctypes signature -- and therefore include/grx.h -- declares."""
    import ast
    from graphrole_amd import _lib
    root = os.path.join(ROOT, 'graphrole_amd')
    checked = 0
    for dirpath, _, files in os.walk(root):
        for fn in files:
            if not fn.endswith('.py'):
                continue
            path = os.path.join(dirpath, fn)
            for node in ast.walk(ast.parse(open(path).read())):
                if not isinstance(node, ast.Call) or any(isinstance(a, ast.Starred) for a in node.args):
                    continue
                f = node.func
                if isinstance(f, ast.Attribute) and f.attr == 'call' and node.args and isinstance(node.args[0], ast.Constant) \
                        and str(node.args[0].value).startswith('grx_'):
                    name, got = node.args[0].value, len(node.args) - 1
                elif isinstance(f, ast.Attribute) and f.attr.startswith('grx_') and f.attr in _lib._SIGNATURES:
                    name, got = f.attr, len(node.args)
                else:
                    continue
                want = len(_lib._SIGNATURES[name][1])
                assert got == want, f'{path}:{node.lineno}: {name} takes {want} arguments, {got} given'
                checked += 1
    assert checked > 60


def test_packed_layout_placement_needs_no_device():
    """grx_packed_row_bytes is host arithmetic: fields fill word 0, then word 1, none across a word, the neighbour count
    last and at most 31 bits wide"""
    from graphrole_amd import _lib
    lib = _lib.load()

    def row_bytes(field_bits, degree_bits, n_out=1):
        L = _lib.PackedLayout()
        L.n_fields, L.degree_bits, L.n_out = len(field_bits), degree_bits, n_out
        for k, b in enumerate(field_bits):
            L.field_bits[k] = b
        return lib.grx_packed_row_bytes(ctypes.byref(L))

    assert row_bytes([14, 15, 20], 0) == 8                      # BA 1 M, generation 1
    assert row_bytes([20, 20, 28], 14) == 16                    # BA 1 M, generation 2: 68 + 14 bits
    assert row_bytes([32, 32], 0) == 8 and row_bytes([32, 32], 1) == 16
    assert row_bytes([40, 30, 30, 30], 4) == 0                  # 40 | 30 + 30 | 30 + 4: the third word does not exist
    assert row_bytes([62, 62], 31) == 0 and row_bytes([63], 0) == 0 and row_bytes([0], 0) == 0
    assert row_bytes([10], 32) == 0                             # neighbour counts are below 2^31
    assert row_bytes([10] * 8, 0) == 0 and row_bytes([9] * 7, 0, n_out=9) == 0
