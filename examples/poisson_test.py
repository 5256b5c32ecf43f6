#!/usr/bin/env python3
"""Python-3 counterpart of the reference's examples/poisson_test.py:50-124 on the MI355X modules:
L = poisson2d(n); A = L.to_csr(); S = L.to_sss(); pcg(M, b = 1, x0 = 0, 1e-8, 2000) for
M in {S, A, L}; prints info / iter / relres, ||x|| and ||b - A x||.  (The reference script
passes an uninitialised x0; zeros are used here.)  The SSOR and JDSYM parts are out of scope."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse.sparse import spmatrix  # noqa: E402
from pysparse.itsolvers.krylov import pcg  # noqa: E402


def poisson2d(n):  # pysparse/tools/poisson.py:22-37
    L = spmatrix.ll_mat(n * n, n * n, 5 * n * n - 4 * n)
    for i in range(n):
        for j in range(n):
            k = i + n * j
            L[k, k] = 4
            if i > 0:
                L[k, k - 1] = -1
            if i < n - 1:
                L[k, k + 1] = -1
            if j > 0:
                L[k, k - n] = -1
            if j < n - 1:
                L[k, k + n] = -1
    return L


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    tol = 1e-8
    t1 = time.perf_counter()
    L = poisson2d(n)
    print("Time for constructing the matrix using poisson2d        : %8.2f sec" % (time.perf_counter() - t1))
    A, S = L.to_csr(), L.to_sss()
    print(L.nnz, S.nnz, A.nnz)
    b = np.ones(n * n, "d")
    for name, M in (("SSS", S), ("CSR", A), ("LL", L)):
        t1 = time.perf_counter()
        x = np.zeros(n * n, "d")
        info, it, relres = pcg(M, b, x, tol, 2000)
        print("info=%d, iter=%d, relres=%e" % (info, it, relres))
        print("Solve time using %s matrix: %8.2f s" % (name, time.perf_counter() - t1))
        print("norm(x) = %g" % np.linalg.norm(x))
        r = np.empty(n * n, "d")
        M.matvec(x, r)
        print("norm(b - A*x) = %g" % np.linalg.norm(b - r))
