"""
Minimum-description-length costs for RolX model selection (reference:
graphrole/roles/description_length.py) for callers that hold the encoded factors as HOST arrays.
RoleExtractor's grid search computes the same two costs in HBM (roles/extract.py::_select_model: code-book
sizes from the quantiser, masked KL error by grx_nmf_kl_cost -- SURVEY.md section 8f rank 2) and does not
come through here; tests/test_gpu_rolx.py checks the device costs against these formulas.
"""
from typing import Tuple

import numpy as np

from graphrole_amd.types import FactorTuple, MatrixLike


def get_description_length_costs(V: MatrixLike, model: FactorTuple) -> Tuple[float, float]:
    """
    (encoding cost, error cost) of the encoded factor pair for the feature matrix V
    (description_length.py:8-29)
    """
    G_encoded, F_encoded = model
    V_orig = V.values if hasattr(V, 'values') else V
    return get_encoding_cost(model), get_error_cost(V_orig, G_encoded @ F_encoded)


def get_encoding_cost(model: FactorTuple) -> float:
    """bits per value (from the larger codebook) times the number of stored values (:32-41)."""
    G_encoded, F_encoded = model
    codebook = max(len(np.unique(G_encoded)), len(np.unique(F_encoded)))
    return np.ceil(np.log2(codebook)) * (G_encoded.size + F_encoded.size)


def get_error_cost(V: np.ndarray, V_approx: np.ndarray) -> float:
    """Generalised KL divergence of V from its reconstruction, zero entries of V masked (:44-61)."""
    orig = np.asarray(V, dtype=np.float64).ravel()
    approx = np.asarray(V_approx, dtype=np.float64).ravel()
    nonzero = orig != 0
    logs = np.zeros(orig.shape)
    np.log(orig / approx, where=nonzero, out=logs)
    return np.sum(np.where(nonzero, orig * logs - orig + approx, 0))
