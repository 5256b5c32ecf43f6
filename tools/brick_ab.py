#!/usr/bin/env python3
"""The brick form of the single-kernel PCG loop (3-D grid operators; psp_mid.hip) against what runs without it -- psp_coop.hip's
one-row-per-thread loop up to 2^18 rows, the launch-per-phase loops beyond -- in ONE process on the same operator and vectors
(PSP_BRICK_MIN is read per solve): microseconds per iteration from two truncated solves.  Start with PSP_TUNING=1.
Usage: brick_ab.py [pcg|minres] [nx,ny,nz ...]"""
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
argv = sys.argv[1:]
solver = argv.pop(0) if argv and argv[0] in ("pcg", "minres") else "pcg"
fn = L.psp_pcg_dev if solver == "pcg" else L.psp_minres_dev
for g in argv or ["32,32,32", "48,48,48", "64,64,64", "80,80,80", "96,96,96", "100,100,100", "128,64,64"]:
    grid = tuple(int(t) for t in g.split(","))
    A = dev.DeviceCSR.poisson(*grid)
    n = A.shape[0]
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    xb.upload(np.random.default_rng(1).standard_normal(n))
    A.matvec_dev(xb.ptr, bb.ptr)
    check(L.psp_synchronize())
    k1, k2 = 15, 75
    rec = {"brick": [], "other": []}
    res = {}
    for rnd in range(3):
        for mode in ("other", "brick"):
            os.environ["PSP_BRICK_MIN"] = "1" if mode == "brick" else str(1 << 30)
            ts = {}
            for kk in (k1, k1, k2):
                xb.zero()
                info, it, rr = C.c_int(), C.c_int(), C.c_double()
                check(L.psp_synchronize())
                t = time.perf_counter()
                check(fn(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it), C.byref(rr), None))
                check(L.psp_synchronize())
                ts[kk] = time.perf_counter() - t
                assert it.value == (kk + 1 if solver == "pcg" else kk), (it.value, info.value)
            rec[mode].append((ts[k2] - ts[k1]) / (k2 - k1) * 1e6)
            res[mode] = (rr.value, float(np.abs(xb.download()).max()))
    s, f = C.c_longlong(), C.c_longlong()
    L.psp_debug_brick_count(C.byref(s), C.byref(f))
    print("x".join(str(v) for v in grid), solver, json.dumps({
        "n": n, "us_per_iter_bricks": min(rec["brick"]), "us_per_iter_without": min(rec["other"]),
        "speedup": min(rec["other"]) / min(rec["brick"]), "relres_rel_diff": abs(res["brick"][0] - res["other"][0]) / res["other"][0],
        "brick_solves_so_far": s.value, "fallbacks": f.value}), flush=True)
    del aop, kop, K
    A.close()
    bb.free()
    xb.free()
