#!/usr/bin/env python3
"""PCG lazy loop with / without the p update and the pending x update folded into the product (PSP_PCG_LAZYPF; VERDICT r4
'Next' #7), in ONE process on the same operator, alternated: iterations/s and the bits of x.  Start with PSP_TUNING=1.
Usage: lazypf_ab.py [nx,ny,nz ...]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("PSP_TUNING") == "1", "start with PSP_TUNING=1"
from pysparse_amd import _capi, device as dev  # noqa: E402

L = _capi.lib()
check = _capi.check


def run(grid, iters, form):
    A = (dev.DeviceCSR if form == "csr" else dev.DeviceSSS).poisson(*grid)
    n = A.shape[0] if form == "csr" else A.n
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    ones = np.ones(1 << 24)
    for k in range(0, n, ones.size):
        check(L.psp_memcpy_h2d(xb.ptr + 8 * k, ones.ctypes.data, 8 * min(ones.size, n - k)))
    A.matvec_dev(xb.ptr, bb.ptr)
    check(L.psp_synchronize())
    rec = {"0": [], "1": []}
    xs = {}
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ["PSP_PCG_LAZYPF"] = mode
            for kk in (2, iters):
                xb.zero()
                info, it, rr = C.c_int(), C.c_int(), C.c_double()
                check(L.psp_synchronize())
                t = time.perf_counter()
                check(L.psp_pcg_dev(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it), C.byref(rr), None))
                check(L.psp_synchronize())
                dt = time.perf_counter() - t
            rec[mode].append(dt / iters * 1e6)
            xs[mode] = ((info.value, it.value, rr.value), xb.download())
    out = {"us_per_iter_plain": min(rec["0"]), "us_per_iter_folded": min(rec["1"]),
           "gain_pct": 100.0 * (min(rec["0"]) / min(rec["1"]) - 1.0),
           "same_bits": bool(xs["0"][0] == xs["1"][0] and np.array_equal(xs["0"][1], xs["1"][1])), "all_us": rec}
    print("%s %s pcg" % ("x".join(str(g) for g in grid if g), form), json.dumps(out), flush=True)


if __name__ == "__main__":
    grids = sys.argv[1:] or ["512,512,512", "4096,4096,0", "256,256,256", "2048,2048,0", "1024,1024,0"]
    os.environ["PSP_PCG_LAZYX"] = "2"  # the lazy loop at every size (the eager band has its own folded form)
    for g in grids:
        grid = tuple(int(t) for t in g.split(","))
        n = grid[0] * grid[1] * max(grid[2], 1)
        iters = max(40, min(2000, int(4e9 / (138 * n))))
        for form in ("csr", "sss"):
            run(grid, iters, form)
