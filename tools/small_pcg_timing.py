"""Jacobi-PCG microseconds per iteration on small 2-D Poisson problems (host-pointer API, so the
PCIe copies of b and x are inside): the launch-latency end of the range."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev
for N in (100, 300, 1000, 2048, 4096):
    A = dev.DeviceCSR.poisson(N, N)
    K = dev.DeviceJacobi(A)
    n = A.shape[0]
    b = np.ones(n)
    x = np.zeros(n); dev.pcg(A, b, x, 0.0, 50, K)
    k = 2000 if N <= 1000 else 400
    t = time.perf_counter(); x = np.zeros(n); r = dev.pcg(A, b, x, 0.0, k, K); dt = time.perf_counter() - t
    print("poisson2d(%d): %s in %.1f ms = %.1f us/iteration (incl. PCIe of b, x)" % (N, r[:2], dt * 1e3, dt * 1e6 / k), flush=True)
