"""
Sanitizer flavours of the library's HOST code ("host C++ built with -fsanitize=address,undefined").

`make -C graphrole_amd/csrc SAN=asan|tsan` compiles the host side of every translation unit under AddressSanitizer +
UndefinedBehaviorSanitizer resp. ThreadSanitizer (device code as usual) into graphrole_amd/libgrx_asan.so /
libgrx_tsan.so; the host-only test files then run in a subprocess with the matching clang runtime preloaded and
GRX_LIB_PATH pointing at the instrumented build.  Any report makes the sanitizer exit non-zero (halt_on_error).
What is covered: the fork-join pool and the checksum / choice helpers of grx_hostio, the k x F algebra of
grx_host_linalg (eigh, SVD, range finder, NNDSVD plan), the pruner of grx_refex -- the code round 3's advisor found a
real race in.  Kernels and the pinned ring need a GPU and stay outside.
"""
import glob
import os
import subprocess
import sys

import pytest

from tests import util

CSRC = os.path.join(util.ROOT, 'graphrole_amd', 'csrc')
HOST_TESTS = ['tests/test_hostio_cpu.py', 'tests/test_host_linalg_cpu.py', 'tests/test_native_pruner_cpu.py']


def _runtime(name):
    hits = glob.glob(f'/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.{name}-x86_64.so')
    if not hits:
        pytest.skip(f'clang runtime of {name} not installed')
    return sorted(hits)[-1]


def _build(flavour):
    jobs = str(min(8, os.cpu_count() or 1))
    subprocess.check_call(['make', '-C', CSRC, '-j', jobs, f'SAN={flavour}', 'ARCH=gfx950'],
                          stdout=subprocess.DEVNULL)
    lib = os.path.join(util.ROOT, 'graphrole_amd', f'libgrx_{flavour}.so')
    assert os.path.exists(lib)
    return lib


def _run(flavour, runtime, options, tests):
    lib = _build(flavour)
    env = dict(os.environ)
    env.update({'LD_PRELOAD': runtime, 'GRX_LIB_PATH': lib, 'PYTHONDONTWRITEBYTECODE': '1'})
    env.update(options)
    proc = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-p', 'no:cacheprovider'] + tests,
                          cwd=util.ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    tail = proc.stdout[-4000:]
    assert proc.returncode == 0, tail
    assert 'passed' in tail and 'Sanitizer' not in proc.stdout, tail


def test_host_code_is_clean_under_address_and_undefined_behaviour_sanitizers():
    _run('asan', _runtime('asan'),
         {'ASAN_OPTIONS': 'detect_leaks=0:halt_on_error=1:abort_on_error=0',          # CPython itself "leaks" at exit
          'UBSAN_OPTIONS': 'print_stacktrace=1:halt_on_error=1'}, HOST_TESTS)


def test_host_thread_pool_is_clean_under_thread_sanitizer():
    # numpy's bundled OpenBLAS hands work to its own threads through primitives the sanitizer cannot see: one BLAS thread
    # and a suppression for that library (tests/tsan.supp) keep the reports to this repository's code
    supp = os.path.join(util.ROOT, 'tests', 'tsan.supp')
    _run('tsan', _runtime('tsan'),
         {'TSAN_OPTIONS': f'halt_on_error=1:report_signal_unsafe=0:exitcode=66:suppressions={supp}',
          'OPENBLAS_NUM_THREADS': '1', 'OMP_NUM_THREADS': '1'}, HOST_TESTS)
