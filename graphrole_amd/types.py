"""Type aliases shared across graphrole_amd (reference: graphrole/types.py:9-21)."""
from typing import Dict, Tuple, Union

import numpy as np
import pandas as pd

VectorLike = Union[np.ndarray, pd.Series]
MatrixLike = Union[pd.DataFrame, np.ndarray]
DataFrameLike = Union[pd.DataFrame, pd.Series]
Node = Union[int, str]
Edge = Tuple[Node, Node]
DataFrameDict = Dict[str, Dict[Node, float]]
FactorTuple = Tuple[np.ndarray, np.ndarray]
