"""
Seeded synthetic graph generators for the BASELINE.json configurations (numpy, host side).
They produce CSRGraph inputs directly -- a 1 M / 10 M networkx object is never built.

  er_graph(n, m)            G(n, m): m distinct undirected non-loop edges, uniform   (config 2)
  ba_graph(n, m)            Barabasi-Albert preferential attachment (config 3 / 4)
  directed_weighted_graph() directed, weighted U(0.1,5), power-law in-degree, 8 numeric node
                            attributes                                                (config 5)
"""
from __future__ import annotations

import numpy as np

from graphrole_amd.graph.csr import CSRGraph


def er_edges(n: int, m: int, seed: int = 0):
    rng = np.random.default_rng(seed)
    have = np.zeros(0, dtype=np.int64)
    while len(have) < m:
        need = int((m - len(have)) * 1.1) + 16
        a = rng.integers(0, n, size=need)
        b = rng.integers(0, n, size=need)
        keep = a != b
        key = np.minimum(a[keep], b[keep]) * n + np.maximum(a[keep], b[keep])
        have = np.unique(np.concatenate([have, key]))
    have = rng.permutation(have)[:m]
    return have // n, have % n


def er_graph(n: int, m: int, seed: int = 0) -> CSRGraph:
    src, dst = er_edges(n, m, seed)
    return CSRGraph(n, src, dst, validate=False)


def ba_edges(n: int, m: int, seed: int = 0):
    """
    Preferential attachment by the repeated-nodes method: the i-th edge of new node v goes to an
    endpoint drawn uniformly from the endpoint list of all earlier edges.  Vectorised: endpoint
    slot 2k holds the (known) source of edge k, slot 2k+1 its target; a target that points at
    another target slot is resolved by pointer jumping.  A new node's m targets are distinct
    (like networkx.barabasi_albert_graph): a repeated target is drawn again, so the graph has
    exactly (n - m) * m edges.
    """
    rng = np.random.default_rng(seed)
    nodes = np.arange(m, n, dtype=np.int64)
    src = np.repeat(nodes, m)
    M = len(src)
    k = np.arange(M, dtype=np.int64)
    first = k < m                                      # node m attaches to 0..m-1
    limit = 2 * (k // m) * m                           # endpoint slots that exist when node v arrives
    ptr = (rng.random(M) * np.maximum(limit, 1)).astype(np.int64)
    dst = np.where(first, k, -1)

    def resolve(pending):
        # even slot 2j -> src[j]; odd slot 2j+1 -> dst[j]
        cur = ptr.copy()
        while pending.any():
            idx = np.nonzero(pending)[0]
            c = cur[idx]
            even = (c & 1) == 0
            j = c >> 1
            done_even = idx[even]
            dst[done_even] = src[j[even]]
            pending[done_even] = False
            odd_idx = idx[~even]
            jo = j[~even]
            known = dst[jo] >= 0
            dst[odd_idx[known]] = dst[jo[known]]
            pending[odd_idx[known]] = False
            cur[odd_idx[~known]] = cur[jo[~known]]       # jump to what that slot is waiting for

    resolve(~first)
    while True:
        # repeated targets of one new node (all but the first occurrence) are drawn again
        tgt = dst.reshape(-1, m)
        order = np.argsort(tgt, axis=1, kind='stable')
        srt = np.take_along_axis(tgt, order, axis=1)
        rows, pos = np.nonzero(srt[:, 1:] == srt[:, :-1])
        if not len(rows):
            break
        dup = np.zeros(M, dtype=bool)
        dup[rows * m + order[rows, pos + 1]] = True
        idx = np.nonzero(dup)[0]
        ptr[idx] = (rng.random(len(idx)) * np.maximum(limit[idx], 1)).astype(np.int64)
        # later edges may have copied a redrawn target: they keep the value they copied (any
        # endpoint that was a valid draw stays one), only the repeated edges themselves change
        dst[idx] = -1
        resolve(dup)
    return src, dst


def ba_graph(n: int, m: int, seed: int = 0) -> CSRGraph:
    src, dst = ba_edges(n, m, seed)
    return CSRGraph(n, src, dst, validate=False)


def directed_weighted_graph(n: int, m: int, seed: int = 0, n_attr: int = 8) -> CSRGraph:
    """m arcs: sources uniform, targets by preferential attachment on in-degree (power law),
    parallel arcs merged by weight sum; weights U(0.1, 5); attributes: 4 x U(0,1), 2 x Exp(1),
    1 x Pareto(2), 1 x Poisson(3)."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, n, size=m)
    # target of arc k: with prob 1/2 a uniform node, else the target of a uniformly chosen
    # earlier arc (copy model -> power-law in-degree); resolved by pointer jumping
    dst = rng.integers(0, n, size=m)
    copy = rng.random(m) < 0.5
    copy[0] = False
    ref = (rng.random(m) * np.arange(m)).astype(np.int64)
    order = np.nonzero(copy)[0]
    # chains are short (geometric); iterate until stable
    tgt = np.where(copy, -1, dst)
    pending = copy.copy()
    cur = ref.copy()
    while pending.any():
        idx = np.nonzero(pending)[0]
        c = cur[idx]
        known = tgt[c] >= 0
        tgt[idx[known]] = tgt[c[known]]
        pending[idx[known]] = False
        cur[idx[~known]] = cur[c[~known]]
    del order
    w = rng.uniform(0.1, 5.0, size=m)
    key = src * n + tgt
    uniq, inv = np.unique(key, return_inverse=True)
    wsum = np.bincount(inv, weights=w, minlength=len(uniq))
    attrs = {}
    for i in range(n_attr):
        if i < 4:
            attrs[f'u{i}'] = rng.random(n)
        elif i < 6:
            attrs[f'e{i}'] = rng.exponential(1.0, n)
        elif i == 6:
            attrs['pareto'] = rng.pareto(2.0, n)
        else:
            attrs[f'poisson{i}'] = rng.poisson(3.0, n).astype(np.float64)
    return CSRGraph(n, uniq // n, uniq % n, weights=wsum, directed=True, attributes=attrs, validate=False)


# ------------------------------------------------------------------ one generation, many processes
def save_graph(g: CSRGraph, directory: str) -> None:
    """The edge arrays a CSRGraph was built from (+ weights, + attribute columns) as .npy files: what
    ``load_graph`` in another process maps back without running the generator again."""
    import json
    import os
    src, dst, w = g.edge_arrays()
    np.save(os.path.join(directory, 'src.npy'), src)
    np.save(os.path.join(directory, 'dst.npy'), dst)
    if w is not None:
        np.save(os.path.join(directory, 'w.npy'), w)
    for i, values in enumerate(g.attributes.values()):
        np.save(os.path.join(directory, f'attr{i}.npy'), np.asarray(values))
    meta = {'n': g.n, 'directed': g.directed, 'weighted': w is not None, 'integral': g.integral,
            'attributes': list(g.attributes)}
    with open(os.path.join(directory, 'meta.json'), 'w') as fh:
        json.dump(meta, fh)


def load_graph(directory: str) -> CSRGraph:
    """CSRGraph over memory-mapped copies of the arrays ``save_graph`` wrote (read-only, shared page cache)."""
    import json
    import os
    meta = json.load(open(os.path.join(directory, 'meta.json')))
    src = np.load(os.path.join(directory, 'src.npy'), mmap_mode='r')
    dst = np.load(os.path.join(directory, 'dst.npy'), mmap_mode='r')
    w = np.load(os.path.join(directory, 'w.npy'), mmap_mode='r') if meta['weighted'] else None
    attrs = {name: np.load(os.path.join(directory, f'attr{i}.npy'), mmap_mode='r')
             for i, name in enumerate(meta['attributes'])}
    return CSRGraph(meta['n'], src, dst, weights=w, directed=meta['directed'], attributes=attrs or None, validate=False)
