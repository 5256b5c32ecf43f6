import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev
from pysparse_amd._capi import check, lib
from tools.spmv_sweep import time_launches
L = lib()
n = 512 ** 3
rng = np.random.default_rng(0)
bufs = [dev.DeviceBuffer.from_host(rng.standard_normal(n)) for _ in range(5)]
out = dev.DeviceBuffer(8)
p, q, dinv, x, r = (b.ptr for b in bufs)
f_xr = lambda: check(L.psp_k_xr_update(n, 1e-3, p, q, dinv, x, r, out.ptr))
f_pu = lambda: check(L.psp_k_pupdate(n, r, dinv, 0.5, 0, p))
f_dot = lambda: check(L.psp_k_dot(n, p, q, out.ptr))
for name, f, nbytes in (("xr_update(+fold+finish)", f_xr, 56 * n), ("pupdate", f_pu, 32 * n), ("dot(+fold+finish)", f_dot, 16 * n)):
    time_launches(f, 5)
    t = min(time_launches(f, 20) for _ in range(3))
    print("%-26s %.3f ms  %.0f GB/s" % (name, t, nbytes / t / 1e6))
f24 = lambda: check(L.psp_k_pupdate(n, r, None, 0.5, 0, p))
time_launches(f24, 5)
t = min(time_launches(f24, 20) for _ in range(3))
print("%-26s %.3f ms  %.0f GB/s" % ("24n kernel (2r+1w)", t, 24 * n / t / 1e6))
fres = lambda: check(L.psp_k_residual(n, q, r, dinv, out.ptr))
time_launches(fres, 5)
t = min(time_launches(fres, 20) for _ in range(3))
print("%-26s %.3f ms  %.0f GB/s" % ("residual 32n (3r+1w)+fold", t, 32 * n / t / 1e6))
