"""Vectorised Poisson builders: whole diagonals go in through ll_mat.put(values, rows, cols)
(pysparse/tools/poisson_vec.py; LLMat_put, ll_mat.c:2497-2752) instead of one L[i, j] = v per entry."""
import numpy as np

from ..sparse import spmatrix

__all__ = ["poisson1d_vec", "poisson1d_sym_vec", "poisson2d_vec", "poisson2d_sym_vec", "poisson3d_vec",
           "poisson3d_sym_vec"]


def _put_diagonals(L, dims, sym):
    strides = [1]
    for d in dims[:-1]:
        strides.append(strides[-1] * d)
    total = strides[-1] * dims[-1]
    k = np.arange(total, dtype=np.int64)
    L.put(np.full(total, 2.0 * len(dims)), k, k)
    for extent, stride in zip(dims, strides):
        has_lower = k[(k // stride) % extent > 0]  # rows with a neighbour at k - stride
        L.put(np.full(has_lower.size, -1.0), has_lower, has_lower - stride)
        if not sym:
            L.put(np.full(has_lower.size, -1.0), has_lower - stride, has_lower)
    return L


def poisson1d_vec(n):
    return _put_diagonals(spmatrix.ll_mat(n, n, 3 * n - 2), (n,), False)


def poisson1d_sym_vec(n):
    return _put_diagonals(spmatrix.ll_mat_sym(n, 2 * n - 1), (n,), True)


def poisson2d_vec(n):
    n2 = n * n
    return _put_diagonals(spmatrix.ll_mat(n2, n2, 5 * n2 - 4 * n), (n, n), False)


def poisson2d_sym_vec(n):
    n2 = n * n
    return _put_diagonals(spmatrix.ll_mat_sym(n2, 3 * n2 - 2 * n), (n, n), True)


def poisson3d_vec(n):
    n3 = n * n * n
    return _put_diagonals(spmatrix.ll_mat(n3, n3, 7 * n3 - 6 * n * n), (n, n, n), False)


def poisson3d_sym_vec(n):
    n3 = n * n * n
    return _put_diagonals(spmatrix.ll_mat_sym(n3, 4 * n3 - 3 * n * n), (n, n, n), True)
