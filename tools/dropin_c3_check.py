#!/usr/bin/env python3
"""One-off: the drop-in modules at configs[1]'s size (512^3) end to end with host NumPy vectors: poisson_csr / poisson_sss,
csr_from_arrays / sss_from_arrays round trips, matvec, precon.jacobi / precon.ssor, krylov.pcg / minres with a few
iterations.  Looks for size-dependent failures in the glue (32-bit products, launch limits), not for speed."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse.sparse import spmatrix
from pysparse.precon import precon
from pysparse.itsolvers import krylov
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t0 = time.time()
A = spmatrix.poisson_csr(N, N, N); S = spmatrix.poisson_sss(N, N, N)
n = A.shape[0]
print("n", n, "csr nnz", A.nnz, "sss nnz", S.nnz, flush=True)
e = np.ones(n); b = np.empty(n); b2 = np.empty(n)
A.matvec(e, b); S.matvec(e, b2)
assert np.array_equal(b, b2)
ip, ix, dv = A.to_arrays() if hasattr(A, "to_arrays") else (None, None, None)
if ip is not None:
    A2 = spmatrix.csr_from_arrays(ip, ix, dv, (n, n)); y = np.empty(n); A2.matvec(e, y); assert np.array_equal(y, b); del A2
    print("csr round trip ok", flush=True)
sp, si, sv, sd = S.to_arrays()
S2 = spmatrix.sss_from_arrays(sp, si, sv, sd); y = np.empty(n); S2.matvec(e, y); assert np.array_equal(y, b)
print("sss round trip ok", flush=True)
for name, M, K in (("csr jacobi", A, precon.jacobi(A, 1.0, 1)), ("sss ssor", S2, precon.ssor(S2, 1.0, 1)), ("sss none", S, None)):
    for sname, solver in (("pcg", krylov.pcg), ("minres", krylov.minres)):
        x = np.zeros(n)
        t = time.time()
        info, it, rr = solver(M, b, x, 1e-30, 8, K) if K is not None else solver(M, b, x, 1e-30, 8)
        print(name, sname, info, it, "%.3e" % rr, "err %.3e" % np.abs(x - 1).max(), round(time.time() - t, 2), "s", flush=True)
        assert np.isfinite(x).all() and it in (8, 9)
print("ok", round(time.time() - t0, 1), "s")
