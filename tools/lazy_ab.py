#!/usr/bin/env python3
"""Where should the lazy-x PCG loop (pcg_async_loop_lazy: 8 n bytes less per iteration, one more scalar step) take over from
the eager device loop?  Round 1 put the threshold at n = 2^25 when an iteration of either was 9-10 launches; with one-launch
reductions (round 4) the lazy loop is 5 launches and the eager one 6.  In-process A/B on the same buffers: PSP_PCG_LAZYX=0
(eager) against 2 (lazy forced), Jacobi-PCG, device-resident vectors, best of three rounds."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

os.environ["PSP_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

L = lib()
grids = [tuple(int(t) for t in g.split(",")) for g in sys.argv[1:]] or [
    (512, 512, 0), (1024, 1024, 0), (2048, 2048, 0), (128, 128, 128), (4096, 4096, 0), (256, 256, 256), (320, 320, 320), (512, 512, 512)]
for grid in grids:
    A = dev.DeviceCSR.poisson(*grid)
    n = A.shape[0]
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    ones = dev.DeviceBuffer.from_host(np.ones(n))
    b = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, b.ptr)
    del ones
    x = dev.DeviceBuffer(n)
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    iters = 100 if n >= (1 << 24) else 400
    best = {"0": 1e9, "2": 1e9}
    res = {}
    for rnd in range(4):
        for mode in ("2", "0"):
            os.environ["PSP_PCG_LAZYX"] = mode
            x.zero()
            check(L.psp_synchronize())
            t = time.perf_counter()
            check(L.psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, iters, C.byref(info), C.byref(it), C.byref(rr), None))
            check(L.psp_synchronize())
            dt = time.perf_counter() - t
            if rnd:
                best[mode] = min(best[mode], dt)
            res[mode] = (info.value, it.value, rr.value)
    print(json.dumps({"grid": grid, "n": n, "log2n": round(float(np.log2(n)), 2), "lazy_us_per_iter": round(best["2"] / iters * 1e6, 2),
                      "eager_us_per_iter": round(best["0"] / iters * 1e6, 2), "lazy_gain_pct": round(100 * (best["0"] / best["2"] - 1), 2),
                      "same_result": res["0"] == res["2"]}), flush=True)
    del A, K, aop, kop, b, x
