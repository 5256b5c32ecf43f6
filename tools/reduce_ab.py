#!/usr/bin/env python3
"""A/B of the lazy PCG loop's merged reduction (the scan of the px pass reduced together with p.q behind the product:
5 launches per iteration instead of 7 up to n = 2^25, 7 instead of 9 beyond) against the scan's own reduction in front
of the product, INSIDE one process (fresh processes of the same launch differ by up to 8 %):
Jacobi-PCG and Jacobi-MINRES iterations/s with device-resident vectors, the switch PSP_PCG_MERGE_XPQ alternated between
solves (MINRES has no switch: its two columns show the noise), best of three rounds each; also checks that both forms
return the same bits.  (Round 4 first used this tool for the per-workgroup reduction tails that were backed out:
profiles/r4_reduce_tail_per_workgroup_ab.txt.)
    python tools/reduce_ab.py [grid ...]        e.g. 512,512,512 4096,4096,0"""
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
                    os.environ["PSP_PCG_MERGE_XPQ"] = mode
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
            out[key] = {"merged_it_per_s": round(iters / best["1"], 2), "unmerged_it_per_s": round(iters / best["0"], 2),
                        "gain_pct": round(100.0 * (best["0"] / best["1"] - 1.0), 2), "same_bits": res["1"] == res["0"],
                        "us_per_iter_merged": round(best["1"] / iters * 1e6, 2), "us_per_iter_unmerged": round(best["0"] / iters * 1e6, 2)}
            print(key, json.dumps(out[key]), flush=True)
        del A, K, aop, kop, b, x
print(json.dumps(out))
