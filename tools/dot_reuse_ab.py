#!/usr/bin/env python3
"""A/B of the dot epilogue of csr_spmv_w4 / sss_spmv_w4 inside the PCG / MINRES loops: the operand of p.q (v.Av) is x itself,
so its pair is already in the kernel's registers (round 4: reused) instead of being loaded again (PSP_W4_DOT_RELOAD=1: rounds
1-3).  In-process, the switch alternated between solves on the same buffers, best of three rounds; both must return the
same bits.  gain_pct = how much faster the reuse form is.
    python tools/dot_reuse_ab.py [grid ...]"""
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
grids = [tuple(int(t) for t in g.split(",")) for g in sys.argv[1:]] or [(512, 512, 512), (4096, 4096, 0), (256, 256, 256),
                                                                      (1024, 1024, 0)]
out = {}
for grid in grids:
    for form in ("csr", "sss"):
        A = dev.DeviceCSR.poisson(*grid) if form == "csr" else dev.DeviceSSS.poisson(*grid)
        n = A.shape[0] if form == "csr" else A.n
        K = dev.DeviceJacobi(A)
        aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
        ones = dev.DeviceBuffer.from_host(np.ones(n))
        b = dev.DeviceBuffer(n)
        A.matvec_dev(ones.ptr, b.ptr)
        del ones
        x = dev.DeviceBuffer(n)
        info, it, rr = C.c_int(), C.c_int(), C.c_double()
        iters = 100 if n >= (1 << 24) else 400
        for name, fn in (("pcg", L.psp_pcg_dev), ("minres", L.psp_minres_dev)):
            best = {"1": 1e9, "0": 1e9}
            res = {}
            for rnd in range(4):
                for mode in ("1", "0"):
                    os.environ["PSP_W4_DOT_RELOAD"] = mode
                    x.zero()
                    check(L.psp_synchronize())
                    t = time.perf_counter()
                    check(fn(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, iters, C.byref(info), C.byref(it), C.byref(rr), None))
                    check(L.psp_synchronize())
                    dt = time.perf_counter() - t
                    if rnd:
                        best[mode] = min(best[mode], dt)
                    if rnd == 3:
                        res[mode] = (info.value, it.value, rr.value, float(np.abs(x.download()).sum()))
            key = "%s %s %s" % ("x".join(str(g) for g in grid if g), form, name)
            out[key] = {"reload_it_per_s": round(iters / best["1"], 2), "reuse_it_per_s": round(iters / best["0"], 2),
                        "gain_pct": round(100.0 * (best["1"] / best["0"] - 1.0), 2), "same_bits": res["1"] == res["0"],
                        "us_per_iter_reload": round(best["1"] / iters * 1e6, 2), "us_per_iter_reuse": round(best["0"] / iters * 1e6, 2)}
            print(key, json.dumps(out[key]), flush=True)
        del A, K, aop, kop, b, x
print(json.dumps(out))
