"""
Feature pruning (reference: graphrole/features/prune.py).

Device work (libgrx.so): vertical logarithmic binning of the columns (batched radix sort +
threshold walk + relabel) and the pairwise Chebyshev distance matrix.  Host work: the feature
graph has a few dozen nodes at most, so grouping into connected components and picking the
oldest member per group stay in Python, exactly as the reference does (prune.py:76-139).
"""
from __future__ import annotations

import itertools as it
from typing import Dict, Iterable, Iterator, List, Sequence, Set, TypeVar

import numpy as np

from graphrole_amd.graph.graph import AdjacencyDictGraph
from graphrole_amd.types import DataFrameLike, VectorLike

T = TypeVar('T', int, str)


def _kernels():
    from graphrole_amd import backend
    return backend.get()


def vertical_log_binning(arr: VectorLike, frac: float = 0.5) -> VectorLike:
    """
    Reassign the values of an array to vertical logarithmic bins (prune.py:13-56), computed on
    the GPU by grx_vertical_log_bin.
    :param arr: array-like of numbers
    :param frac: fraction in (0, 1) of the still unbinned values that goes to each next bin
    """
    if not 0 < frac < 1:
        raise ValueError('must specify frac in interval (0, 1)')
    values = np.ascontiguousarray(np.asarray(arr), dtype=np.float64)
    if values.size == 0:
        return np.zeros(0, dtype=int)
    K = _kernels()
    bins, nbins = K.vertical_log_bin(K.to_device(values.reshape(1, -1)), frac)
    if int(K.to_host(nbins)[0]) < 0:
        # uint8 labels / GRX_MAX_BINS thresholds: the pipeline's frac = 0.5 needs < 70 bins for any n < 2^63,
        # only a tiny frac on many distinct values gets here
        raise NotImplementedError(f'vertical_log_binning: more than 128 bins (frac={frac}); the device kernels '
                                  'label bins with 7 bits and graphrole_amd has no CPU fallback')
    return K.to_host(bins)[0].astype(int)


class FeaturePruner:

    """ Determines redundant features to be removed from future recursive aggregations """

    def __init__(self, generation_dict: Dict[int, Iterable[str]], feature_group_thresh: int) -> None:
        """
        :param generation_dict: generation number -> container of the feature names recorded at
          that generation (the reference passes {gen: {name: {node: value}}}; only the names are
          ever consulted, prune.py:125-127)
        :param feature_group_thresh: features whose binned versions differ by at most this
          Chebyshev distance are grouped
        """
        self._generation_dict = generation_dict
        self._feature_group_thresh = feature_group_thresh

    # ------------------------------------------------------------------ reference API
    def prune_features(self, features: DataFrameLike) -> List[str]:
        """Names to drop: every member of a feature group except its oldest (prune.py:76-92)."""
        return self._drop_list(self._group_features(features))

    def _group_features(self, features: DataFrameLike) -> Iterator[Set[str]]:
        """Bin every column, connect columns within the distance threshold, return the
        connected components (prune.py:94-116)."""
        names = list(features.columns)
        if not names:
            return iter(())
        K = _kernels()
        block = K.to_device(np.ascontiguousarray(features.to_numpy(dtype=np.float64).T))
        bins, _ = K.vertical_log_bin(block)
        n = block.shape[1]
        dist = K.to_host(K.chebyshev([bins[j] for j in range(len(names))], n))
        return self.groups_from_distances(names, dist)

    def _get_oldest_feature(self, feature_names: Set[T]) -> T:
        """Member recorded in the earliest generation; ties and unrecorded members by name
        (prune.py:118-130)."""
        for gen in range(len(self._generation_dict)):
            recorded = feature_names.intersection(self._generation_dict[gen])
            if recorded:
                return self._set_getitem(recorded)
        return self._set_getitem(feature_names)

    @staticmethod
    def _set_getitem(s: Set[T]) -> T:
        """Deterministic representative of a set: its minimum (prune.py:132-139)."""
        return min(s)

    # ------------------------------------------------------------------ engine API
    def groups_from_distances(self, names: Sequence[str], dist: np.ndarray) -> Iterator[Set[str]]:
        """Connected components of the graph {(p,q): dist[p,q] <= thresh} (only linked names)."""
        pairs = it.combinations(range(len(names)), 2)
        edges = [(names[p], names[q]) for p, q in pairs if dist[p, q] <= self._feature_group_thresh]
        return AdjacencyDictGraph(edges).get_connected_components()

    def prune_from_distances(self, names: Sequence[str], dist: np.ndarray) -> List[str]:
        """prune_features for columns that are already binned/compared on the device."""
        return self._drop_list(self.groups_from_distances(names, dist))

    def _drop_list(self, groups: Iterable[Set[str]]) -> List[str]:
        drop: List[str] = []
        for group in groups:
            if len(group) == 1:
                continue
            keep = self._get_oldest_feature(group)
            drop.extend(group - {keep})
        return drop
