"""
TEST INFRASTRUCTURE (checker only; the product never imports oracle/).

Restatement of the reference's igraph adapter, graphrole/graph/interface/igraph.py, for graphs WITH self-loops and
parallel edges, in plain Python loops over an edge list.  igraph itself is absent from this image, so the library
behaviour the adapter relies on is ASSUMED from igraph's documentation (parity against igraph: unpinned):

  * ``Edge.tuple``                       (source, target); an undirected edge is reported as (min, max)
  * ``Graph.neighbors(v, mode='out')``   neighbour ids in ascending order, one entry per parallel edge; an
                                         undirected self-loop lists v twice, a directed one once
  * ``Vertex.degree(mode)``              number of edge ends: a self-loop counts twice in the undirected / total
                                         degree, once in each of 'in' and 'out'
  * ``Graph.neighborhood(v, order=1, mode='out')``   the distinct vertices {v} | out-neighbours(v)

Everything above those four calls follows the reference line by line:
  edge_weights dict                     igraph.py:36-39   (one entry per distinct tuple, LAST weight wins)
  local features                        igraph.py:61-76, 129-162
  ego-net features                      igraph.py:78-98, 164-205
  recursion over G.neighbors            features/extract.py:98-119 (reindex with repeated labels: multiset)
"""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import refex


def _tuples(edges, directed: bool) -> List[Tuple[int, int]]:
    return [(int(a), int(b)) if directed else (min(int(a), int(b)), max(int(a), int(b))) for a, b in edges]


def edge_weights(edges, directed: bool, weights) -> Dict[Tuple[int, int], float]:
    """igraph.py:36-39: {edge.tuple: edge.attributes().get('weight', 1)}"""
    out: Dict[Tuple[int, int], float] = {}
    for i, t in enumerate(_tuples(edges, directed)):
        out[t] = 1 if weights is None else weights[i]
    return out


def neighbors(n: int, edges, directed: bool) -> List[List[int]]:
    """Graph.neighbors(v, mode='out') for every v (assumed igraph semantics, see the module docstring)."""
    out: List[List[int]] = [[] for _ in range(n)]
    for a, b in _tuples(edges, directed):
        out[a].append(b)
        if not directed:
            out[b].append(a)                                   # a loop (v, v): v is appended twice
    return [sorted(x) for x in out]


def local_features(n: int, edges, directed: bool, weights) -> Tuple[List[str], np.ndarray]:
    """igraph.py:61-76 with _get_degree_dict / _get_node_degree (:129-162)."""
    tuples = _tuples(edges, directed)
    ew = edge_weights(edges, directed, weights)

    def degree(node: int, mode: Optional[str]):
        if weights is not None:                                # igraph.py:138-142 -> _get_node_degree
            if directed and mode:
                pick = 0 if mode == 'out' else 1               # itemgetter(0) source / itemgetter(1) target
                return sum(w for e, w in ew.items() if node == e[pick])
            return sum(w for e, w in ew.items() if node in e)
        # Vertex.degree(mode): edge ends of the multigraph
        if directed and mode == 'out':
            return sum(1 for a, _ in tuples if a == node)
        if directed and mode == 'in':
            return sum(1 for _, b in tuples if b == node)
        return sum((a == node) + (b == node) for a, b in tuples)

    if directed:
        names = ['in_degree', 'out_degree', 'total_degree']
        cols = [[degree(v, 'in') for v in range(n)], [degree(v, 'out') for v in range(n)],
                [degree(v, None) for v in range(n)]]
    else:
        names = ['degree']
        cols = [[degree(v, None) for v in range(n)]]
    return names, np.array(cols, dtype=np.float64).T


def egonet_features(n: int, edges, directed: bool, weights) -> Tuple[List[str], np.ndarray]:
    """igraph.py:78-98 with _get_edge_sum_from_nodes / _get_edge_boundary / _is_boundary (:164-205)."""
    tuples = _tuples(edges, directed)
    ew = edge_weights(edges, directed, weights)
    nbrs = neighbors(n, edges, directed)
    out = np.zeros((n, 2), dtype=np.float64)
    everyone = set(range(n))
    for v in range(n):
        ego_nodes = sorted({v} | set(nbrs[v]))                 # Graph.neighborhood(v, order=1, mode='out')
        interior = set(ego_nodes)
        exterior = everyone - interior

        def is_boundary(edge):
            v1, v2 = edge
            if directed:
                return v1 in interior and v2 in exterior
            return (v1 in interior and v2 in exterior) or (v1 in exterior and v2 in interior)

        boundary = [t for t in tuples if is_boundary(t)]       # one entry per parallel edge
        out[v, 0] = sum(w for (s, t), w in ew.items() if s in ego_nodes and t in ego_nodes)
        out[v, 1] = sum(w for e, w in ew.items() if e in boundary)   # ... but the dict holds each tuple once
    return ['internal_edges', 'external_edges'], out


def neighborhood_features(n: int, edges, directed: bool, weights) -> Tuple[List[str], np.ndarray]:
    """base.py:18-26: local then ego-net columns."""
    n1, x1 = local_features(n, edges, directed, weights)
    n2, x2 = egonet_features(n, edges, directed, weights)
    return n1 + n2, np.hstack([x1, x2])


def extract_features(n: int, edges, directed: bool, weights=None, max_generations: int = 10,
                     aggs: Sequence[str] = ('sum', 'mean')) -> refex.RefexResult:
    """RecursiveFeatureExtractor over the igraph adapter: generation 0 as above, recursion over the neighbour
    multiset of Graph.neighbors (features/extract.py:98-119)."""
    lists = neighbors(n, edges, directed)
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum([len(x) for x in lists], out=row_ptr[1:])
    col = np.array([j for x in lists for j in x], dtype=np.int32)
    og = refex.OracleGraph(labels=list(range(n)), row_ptr=row_ptr, col=col, w=None, directed=directed,
                           num_edges=len(edges), adj_col=col.copy())
    return refex.extract_features(og, max_generations=max_generations, fast=False,
                                  gen0=neighborhood_features(n, edges, directed, weights), aggs=aggs)
