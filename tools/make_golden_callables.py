#!/usr/bin/env python3
"""
tools/make_golden_callables.py -- fixtures for user-CALLABLE aggregations (tests/golden/refex_callable_<name>.npz) by
RUNNING THE REFERENCE's RecursiveFeatureExtractor with aggs lists that hold plain Python functions
(graphrole/features/extract.py:26,111: the list goes straight to DataFrame.agg).  The functions live in
tests/graphs.py (no reference code); the reference is imported here and only here; the fixtures are data: the graph
(edges, weights, adjacency order, attribute kwargs), the final table (columns, values, dtypes), generation count.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_callables.py
"""
import json
import os
import sys
import warnings

import numpy as np

REF = '/root/reference'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, 'examples'))
sys.path.insert(0, ROOT)
warnings.simplefilter('ignore')

from graphrole import RecursiveFeatureExtractor              # noqa: E402
from tests import graphs as G_                                # noqa: E402
from tools.make_golden import REFEX_CASES, adjacency_arrays, graph_arrays   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def main():
    for name, (graph, spec, max_generations) in G_.CALLABLE_CASES.items():
        G, kwargs = REFEX_CASES[graph]()
        aggs = G_.resolve_aggs(spec)
        fe = RecursiveFeatureExtractor(G, max_generations=max_generations, aggs=aggs, **kwargs)
        final = fe.extract_features()
        labels, src, dst, w = graph_arrays(G)
        adj_ptr, adj_idx = adjacency_arrays(G, labels)
        ordered = final.loc[labels]
        np.savez_compressed(
            os.path.join(OUT, f'refex_callable_{name}.npz'), n=len(labels), src=src, dst=dst, w=w, directed=G.is_directed(),
            labels_json=json.dumps(labels), kwargs_json=json.dumps(kwargs), num_edges=G.number_of_edges(),
            max_generations=max_generations, aggs_json=json.dumps(spec), adj_ptr=adj_ptr, adj_idx=adj_idx,
            generation_count=fe.generation_count, final_columns_json=json.dumps(list(final.columns)),
            final_values=ordered.values.astype(np.float64), final_dtypes_json=json.dumps([str(t) for t in final.dtypes]))
        print(f'refex_callable_{name}: final {final.shape}, generations {fe.generation_count}, dtypes {sorted(set(str(t) for t in final.dtypes))}')


if __name__ == '__main__':
    main()
