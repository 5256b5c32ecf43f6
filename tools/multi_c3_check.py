#!/usr/bin/env python3
"""One-off: the device-list route of the drop-in modules at 512^3 with host arrays (two and four ranks sharing device 0):
csr_from_arrays(..., devices=[...]) against poisson_csr(..., devices=[...]) and the single-device operator."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse.sparse import spmatrix
from pysparse.precon import precon
from pysparse.itsolvers import krylov
from oracle import oracle as O
N = 512
A1 = spmatrix.poisson_csr(N, N, N)
n = A1.shape[0]
x = np.random.default_rng(1).standard_normal(n); y1 = np.empty(n)
A1.matvec(x, y1)
H = O.poisson_csr(N, N, N)
for devs in ([0, 0], [0, 0, 0, 0]):
    t = time.time()
    A = spmatrix.csr_from_arrays(H.ind, H.col, H.val, (n, n), devices=devs)
    y = np.empty(n); A.matvec(x, y)
    print("from_arrays devices", devs, "equal", np.array_equal(y, y1), round(time.time() - t, 1), "s", flush=True)
    K = precon.jacobi(A, 1.0, 1)
    b = np.empty(n); A.matvec(np.ones(n), b)
    xs = np.zeros(n); info, it, rr = krylov.pcg(A, b, xs, 1e-30, 5, K)
    print("  pcg", info, it, "%.3e" % rr, flush=True)
    del A, K
    G = spmatrix.poisson_csr(N, N, N, devices=devs)
    y = np.empty(n); G.matvec(x, y)
    print("poisson devices", devs, "equal", np.array_equal(y, y1), flush=True)
    del G
