"""
Annotations of the host-side classes.  The names are the ones a GraphRole user may already import
(graphrole/types.py); values that live in HBM are torch tensors and are annotated where they occur.
"""
from __future__ import annotations

import typing as t

import numpy as np
import pandas as pd

#: label of a node, exactly as the graph library reports it
Node = t.Union[int, str]

#: (node-role factor N x r, role-feature factor r x F) as host arrays
FactorTuple = t.Tuple[np.ndarray, np.ndarray]

#: what the public methods accept / return for tables, matrices and single columns
DataFrameLike = t.Union[pd.DataFrame, pd.Series]
MatrixLike = t.Union[np.ndarray, pd.DataFrame]
VectorLike = t.Union[pd.Series, np.ndarray]

#: column name -> {node label -> value}: the shape of DataFrame.to_dict()
DataFrameDict = t.Dict[str, t.Dict[Node, float]]
