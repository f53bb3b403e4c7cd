"""
Adapter registry (reference: graphrole/graph/interface/__init__.py:12-53): the adapter is looked
up by the root package of ``G.__module__``.  ``graphrole_amd`` is an added key for CSRGraph.
The igraph adapter never imports igraph (absent from this image): it uses the Graph object's
public methods only, so the key is always registered.
"""
from typing import List, Optional

from graphrole_amd.graph.interface.base import BaseGraphInterface, DeviceGraphInterface  # noqa: F401
from graphrole_amd.graph.interface.csr import CSRInterface
from graphrole_amd.graph.interface.igraph import IgraphInterface
from graphrole_amd.graph.interface.networkx import NetworkxInterface

INTERFACES = {
    'networkx': NetworkxInterface,
    'igraph': IgraphInterface,
    'graphrole_amd': CSRInterface,
}


def get_supported_graph_libraries() -> List[str]:
    return list(INTERFACES.keys())


def get_interface(G) -> Optional[type]:
    """Adapter class for G, or None when G is not from a supported library (:39-53)."""
    try:
        package = G.__module__.split('.')[0]
    except (AttributeError, IndexError):
        return None
    return INTERFACES.get(package)
