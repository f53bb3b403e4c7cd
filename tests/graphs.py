"""
Deterministic networkx graph builders shared by tools/make_golden.py (which runs the reference on
them) and the tests (which run graphrole_amd on them).  No reference code involved.
"""
import networkx as nx
import numpy as np


def er(n, m, seed):
    return nx.gnm_random_graph(n, m, seed=seed), {}


def ba(n, m, seed):
    return nx.barabasi_albert_graph(n, m, seed=seed), {}


def directed_weighted_attrs(n=200, m=900, seed=3):
    G = nx.gnm_random_graph(n, m, seed=seed, directed=True)
    rng = np.random.default_rng(seed)
    for u, v in G.edges:
        G[u][v]['weight'] = float(rng.uniform(0.1, 5.0))
    G.add_edge(5, 5, weight=2.5)
    G.add_edge(7, 7, weight=1.25)
    for node in G.nodes:
        G.nodes[node]['a_uniform'] = float(rng.random())
        G.nodes[node]['a_poisson'] = int(rng.poisson(3))
        G.nodes[node]['a_text'] = 'not numeric'
        if node % 3 == 0:
            G.nodes[node]['a_sparse'] = float(rng.exponential(1.0))
    return G, {'attributes': True}


def loops_dangling(seed=4):
    G = nx.gnm_random_graph(150, 400, seed=seed)
    G.add_edge(3, 3)
    G.add_edge(9, 9)
    G.add_nodes_from([1000, 1001])
    return G, {}


def directed_unweighted(seed=6):
    G = nx.gnm_random_graph(120, 500, seed=seed, directed=True)
    G.add_edge(2, 2)
    return G, {}


def path4():
    return nx.Graph([('a', 'b'), ('a', 'c'), ('c', 'd')]), {}


IFACE7_EDGES = [(0, 1), (0, 2), (0, 3), (3, 6), (4, 5), (4, 6), (5, 6)]
IFACE7_WEIGHTS = [2, 1.5, 3, 0.25, 0.75, 2.5, 1]
IFACE7_ATTRS = {
    0: {'attr1': 1.00, 'attr2': 0.00},
    1: {'attr2': 1.00},
    2: {'attr2': 2.00},
    3: {'attr2': 3.00},
    4: {'attr2': 4.00},
    5: {'attr2': 5.00},
    6: {'attr2': 6.00},
}


def iface7():
    return nx.Graph(IFACE7_EDGES), {}


def iface7_directed_weighted():
    G = nx.DiGraph()
    for e, w in zip(IFACE7_EDGES, IFACE7_WEIGHTS):
        G.add_edge(*e, weight=w)
    return G, {}


def iface7_attrs():
    G = nx.Graph(IFACE7_EDGES)
    nx.set_node_attributes(G, IFACE7_ATTRS)
    return G


# golden cases that reuse a graph with another aggregation list (extract.py:36-47 `aggs`)
CASE_AGGS = {
    'karate_minmax': ('karate', ['sum', 'mean', 'min', 'max']),
    'ba300_maxsum': ('ba300', ['max', 'sum']),
    'dw200_minmax': ('dw200_attrs', ['min', 'max', 'mean']),
    'loops_dangling150_minmax': ('loops_dangling150', ['sum', 'min', 'max']),
    'ba300_stdvar': ('ba300', ['mean', 'std', 'var']),
    'karate_sumstd': ('karate', ['sum', 'std']),
    # third entry: max_generations (products grow doubly exponentially; the reference wraps int64 beyond 2^63)
    'iface7_prod': ('iface7', ['sum', 'prod'], 4),
    'dw200_prod': ('dw200_attrs', ['prod', 'mean'], 2),
    'path4_prod': ('path4', ['prod', 'max'], 3),
    # round 3: the reference's wrapping int64 products, median, count / size (pandas names; any aggregatable is legal)
    'karate_prodwrap': ('karate', ['prod'], 4),
    'karate_sumprodwrap': ('karate', ['sum', 'prod'], 4),
    'ba300_prodmean': ('ba300', ['prod', 'mean'], 3),
    'karate_summedian': ('karate', ['sum', 'median']),
    'ba300_medianmean': ('ba300', ['median', 'mean'], 4),
    'dw200_medianmax': ('dw200_attrs', ['median', 'max'], 3),
    'loops_dangling150_median': ('loops_dangling150', ['median'], 4),
    'karate_sumcount': ('karate', ['sum', 'count']),
    'er300_meansize': ('er300', ['mean', 'size'], 4),
    'loops_dangling150_countmax': ('loops_dangling150', ['count', 'max'], 4),
    'directed120_prodcount': ('directed120', ['prod', 'count'], 3),
}

BUILDERS = {
    'er300': lambda: er(300, 1500, 1),
    'ba300': lambda: ba(300, 3, 2),
    'dw200_attrs': directed_weighted_attrs,
    'loops_dangling150': loops_dangling,
    'directed120': directed_unweighted,
    'path4': path4,
    'iface7': iface7,
    'iface7_dw': iface7_directed_weighted,
    'er2000': lambda: er(2000, 20000, 0),
    'ba2000': lambda: ba(2000, 10, 0),
}


# ---- aggregation CALLABLES (features/extract.py:26,111: anything DataFrame.agg accepts).  Functions of a Series that
# return a scalar; shared by tools/make_golden_callables.py (the reference runs them) and the tests (graphrole_amd does).
def spread(s):
    return s.max() - s.min()


def second_largest(s):
    values = np.sort(s.to_numpy())
    return float(values[-2]) if len(values) > 1 else np.nan


CALLABLE_CASES = {
    # name: (graph builder name, aggs with callables by registry name, max_generations)
    'karate_sum_spread': ('karate', ['sum', 'callable:spread'], 4),
    'ba300_second_mean': ('ba300', ['callable:second_largest', 'mean'], 3),
    'loops_dangling150_spread': ('loops_dangling150', ['callable:spread'], 3),
    'er300_spread_max': ('er300', ['callable:spread', 'max'], 3),
    # aggregation NAMES pandas knows and the device has no kernel for (extract.py:26,47,111: any name goes to DataFrame.agg)
    'ba300_sum_sem': ('ba300', ['sum', 'sem'], 3),
    'karate_skew_nunique': ('karate', ['skew', 'nunique'], 3),
    'loops_dangling150_sem_max': ('loops_dangling150', ['sem', 'max'], 3),
}
CALLABLES = {'spread': spread, 'second_largest': second_largest}


def resolve_aggs(spec):
    return [CALLABLES[a.split(':', 1)[1]] if isinstance(a, str) and a.startswith('callable:') else a for a in spec]
