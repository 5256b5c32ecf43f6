"""Finite-difference Laplacians as ll_mat objects, natural ordering k = i + n*j (+ n*n*l), Dirichlet
truncation -- the matrices of pysparse/tools/poisson.py:4-66 (1-D, 2-D general / symmetric / block
built) plus the 3-D 7-point operators the MI355X configs use, and direct device constructors for
sizes where an ll_mat is out of the question.

The ll_mat builders fill one entry at a time through L[i, j] = v exactly like the reference (so they
exercise the same sorted-insertion path, ll_mat.c:250-356); use the *_vec versions or the device
constructors for anything large."""
from ..sparse import spmatrix

__all__ = ["poisson1d", "poisson1d_sym", "poisson2d", "poisson2d_sym", "poisson2d_sym_blk",
           "poisson3d", "poisson3d_sym", "poisson2d_csr", "poisson3d_csr", "poisson2d_sss", "poisson3d_sss"]


def _fill(L, dims, sym):
    """dims = (n0, n1, ...): grid extents, fastest index first; diagonal 2*len(dims), neighbours -1"""
    strides = [1]
    for d in dims[:-1]:
        strides.append(strides[-1] * d)
    total = strides[-1] * dims[-1]
    diag = 2 * len(dims)
    for k in range(total):
        L[k, k] = diag
        for extent, stride in zip(dims, strides):
            pos = (k // stride) % extent
            if pos > 0:
                L[k, k - stride] = -1
            if not sym and pos < extent - 1:
                L[k, k + stride] = -1
    return L


def poisson1d(n):
    return _fill(spmatrix.ll_mat(n, n, 3 * n - 2), (n,), False)


def poisson1d_sym(n):
    return _fill(spmatrix.ll_mat_sym(n, 2 * n - 1), (n,), True)


def poisson2d(n):
    n2 = n * n
    return _fill(spmatrix.ll_mat(n2, n2, 5 * n2 - 4 * n), (n, n), False)


def poisson2d_sym(n):
    n2 = n * n
    return _fill(spmatrix.ll_mat_sym(n2, 3 * n2 - 2 * n), (n, n), True)


def poisson2d_sym_blk(n):
    """Same matrix as poisson2d_sym, assembled from n x n blocks by sub-matrix assignment (poisson.py:50-66)."""
    n2 = n * n
    L = spmatrix.ll_mat_sym(n2, 3 * n2 - 2 * n)
    minus_identity = spmatrix.ll_mat_sym(n, n)
    for i in range(n):
        minus_identity[i, i] = -1
    P = spmatrix.ll_mat_sym(n, 2 * n - 1)
    for i in range(n):
        P[i, i] = 4
        if i > 0:
            P[i, i - 1] = -1
    for i in range(0, n2, n):
        L[i:i + n, i:i + n] = P
        if i > 0:
            L[i:i + n, i - n:i] = minus_identity
    return L


def poisson3d(n):
    n3 = n * n * n
    return _fill(spmatrix.ll_mat(n3, n3, 7 * n3 - 6 * n * n), (n, n, n), False)


def poisson3d_sym(n):
    n3 = n * n * n
    return _fill(spmatrix.ll_mat_sym(n3, 4 * n3 - 3 * n * n), (n, n, n), True)


# ---- generated on the GPU, bit-identical to poisson2d(n).to_csr() etc. (tests/test_spmatrix_host.py,
# tests/test_gpu_spmv.py::test_poisson_generator_structure_bit_exact)

def poisson2d_csr(n):
    return spmatrix.poisson_csr(n, n)


def poisson3d_csr(n):
    return spmatrix.poisson_csr(n, n, n)


def poisson2d_sss(n):
    return spmatrix.poisson_sss(n, n)


def poisson3d_sss(n):
    return spmatrix.poisson_sss(n, n, n)
