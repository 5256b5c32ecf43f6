#!/usr/bin/env python3
"""Python-3 counterpart of the reference's examples/demo_pcg.py:23-100 on the MI355X modules:
per MatrixMarket problem, b = A*e and three PCG solves (no preconditioner, Jacobi,
SSOR on A.to_sss()); prints n, nnz, iter, relres, ||x-e||_inf, info, setup and solve seconds.

  python examples/demo_pcg.py problem.mtx [problem2.mtx ...]
  python examples/demo_pcg.py --poisson 100        # generate poisson2d_sym(100) on the fly
"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse.sparse import spmatrix  # noqa: E402
from pysparse.itsolvers.krylov import pcg  # noqa: E402
from pysparse.precon import precon  # noqa: E402


def write_poisson_mtx(n, path):
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write("%d %d %d\n" % (n * n, n * n, 3 * n * n - 2 * n))
        for j in range(n):
            for i in range(n):
                k = i + n * j + 1
                if j > 0:
                    f.write("%d %d -1\n" % (k, k - n))
                if i > 0:
                    f.write("%d %d -1\n" % (k, k - 1))
                f.write("%d %d 4\n" % (k, k))


def test_pcg(problems, tol=1.0e-6):
    head1 = "%10s  %8s  %9s  " % ("Name", "n", "nnz")
    head2 = "%6s  %8s  %8s  %4s  %6s  %6s" % ("iter", "relres", "error", "info", "form M", "solve")
    print("-" * (len(head1) + len(head2)))
    print(head1 + head2)
    print("-" * (len(head1) + len(head2)))
    for problem in problems:
        A = spmatrix.ll_mat_from_mtx(problem)
        (m, n) = A.shape
        if m != n:
            break
        name = os.path.basename(problem)
        if name.endswith(".mtx"):
            name = name[:-4]
        e = np.ones(n, "d")
        b = np.empty(n, "d")
        A.matvec(e, b)
        x = np.zeros(n, "d")
        t = time.perf_counter()
        info, it, relres = pcg(A, b, x, tol, 2 * n)
        t_solve = time.perf_counter() - t
        err = np.linalg.norm(x - e, ord=np.inf)
        print("%10s  %8d  %9d  %6d  %8.1e  %8.1e  %4d  %6.2f  %6.2f" % (name, n, A.nnz, it, relres, err, info, 0.0, t_solve))
        x = np.zeros(n, "d")
        t = time.perf_counter()
        M = precon.jacobi(A, 1.0, 1)
        t_m = time.perf_counter() - t
        t = time.perf_counter()
        info, it, relres = pcg(A, b, x, tol, 2 * n, M)
        t_solve = time.perf_counter() - t
        err = np.linalg.norm(x - e, ord=np.inf)
        print("%10s  %8s  %9s  %6d  %8.1e  %8.1e  %4d  %6.2f  %6.2f" % ("", "", "", it, relres, err, info, t_m, t_solve))
        x = np.zeros(n, "d")
        t = time.perf_counter()
        M = precon.ssor(A.to_sss(), 1.0, 1)  # demo_pcg.py:84-86
        t_m = time.perf_counter() - t
        t = time.perf_counter()
        info, it, relres = pcg(A, b, x, tol, 2 * n, M)
        t_solve = time.perf_counter() - t
        err = np.linalg.norm(x - e, ord=np.inf)
        print("%10s  %8s  %9s  %6d  %8.1e  %8.1e  %4d  %6.2f  %6.2f" % ("", "", "", it, relres, err, info, t_m, t_solve))


if __name__ == "__main__":
    args = sys.argv[1:]
    if len(args) == 2 and args[0] == "--poisson":
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "poi2d_%s.mtx" % args[1])
            write_poisson_mtx(int(args[1]), p)
            test_pcg([p])
    elif args:
        test_pcg(args)
    else:
        sys.stderr.write(__doc__)
        sys.exit(1)
