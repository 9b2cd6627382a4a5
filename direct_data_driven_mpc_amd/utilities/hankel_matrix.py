"""Hankel helpers with the reference's signatures
(direct_data_driven_mpc/utilities/hankel_matrix.py:5,55), computed on the GPU.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

from ..engine import hankel_matrix_batched


def hankel_matrix(X: np.ndarray, L: int) -> np.ndarray:
    """Block-Hankel matrix of `X` (N, n) with `L` block rows -> (L*n, N-L+1).

    Same contract as hankel_matrix.py:5-53: H[k*n + ch, i] = X[i+k, ch];
    ValueError if N < L (hankel_matrix.py:43-44).  The gather runs as a HIP
    kernel (ddmpc_hankel); the result is bit-identical to the reference's copy loop.
    """
    X = np.asarray(X, dtype=np.float64)
    if X.ndim != 2:
        raise ValueError("X must be a 2-D array of shape (N, n)")
    if X.shape[0] < L:
        raise ValueError("N must be greater than or equal to L.")
    return hankel_matrix_batched(X[None], L)[0]


def evaluate_persistent_excitation(X: np.ndarray, order: int) -> Tuple[int, bool]:
    """(rank, is_persistently_exciting) of the order-`order` Hankel matrix of `X`
    (hankel_matrix.py:55-87; rank by SVD with numpy's default tolerance, as the
    reference does -- construction-time validation, not part of the per-step path)."""
    n = X.shape[1]
    rank = int(np.linalg.matrix_rank(hankel_matrix(X, order)))
    return rank, rank == n * order
