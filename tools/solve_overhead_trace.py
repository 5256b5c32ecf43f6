"""30 one-iteration PCG solves at 1024^2 through psp_pcg_dev: the program for `rocprofv3 --hip-trace --stats` (which API calls a
solve is made of: ~12 calls of 2.5-4 us, two stream synchronisations of ~15 us; tools/solve_overhead.py has the totals)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pysparse_amd import _capi, device as dev
L = _capi.lib(); check = _capi.check
A = dev.DeviceCSR.poisson(1024, 1024); n = A.shape[0]; K = dev.DeviceJacobi(A)
aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
xb.upload(np.ones(n)); A.matvec_dev(xb.ptr, bb.ptr); check(L.psp_synchronize())
for _ in range(30):
    xb.zero(); info, it, rr = C.c_int(), C.c_int(), C.c_double()
    check(L.psp_pcg_dev(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, 1, C.byref(info), C.byref(it), C.byref(rr), None))
check(L.psp_synchronize())
