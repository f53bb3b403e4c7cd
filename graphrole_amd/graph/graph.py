"""
Small undirected helper graph used for the *feature* graph of the pruner (a few dozen nodes at
most), host side.  Mirrors the behaviour of the reference's AdjacencyDictGraph
(graphrole/graph/graph.py:7-57): only nodes that occur in at least one edge exist; components
are produced in first-seen order.
"""
from typing import Dict, Hashable, Iterable, Iterator, List, Set, Tuple


class AdjacencyDictGraph:

    def __init__(self, edges: Iterable[Tuple[Hashable, Hashable]]) -> None:
        self.edges = edges
        self.adj_dict: Dict[Hashable, Set[Hashable]] = {}
        for left, right in edges:
            self.adj_dict.setdefault(left, set()).add(right)
            self.adj_dict.setdefault(right, set()).add(left)

    def get_connected_components(self) -> Iterator[Set[Hashable]]:
        done: Set[Hashable] = set()
        for start in self.adj_dict:
            if start in done:
                continue
            members = self._dfs(start)
            done |= members
            yield members

    def _dfs(self, node: Hashable) -> Set[Hashable]:
        reached: Set[Hashable] = set()
        frontier: List[Hashable] = [node]
        while frontier:
            current = frontier.pop()
            if current in reached:
                continue
            reached.add(current)
            frontier.extend(self.adj_dict[current] - reached)
        return reached
