#!/usr/bin/env python3
"""Where should a reduction over per-workgroup partial sums stop being ONE finishing block and become a group fold over the
whole chip + the finishing block?  (Round 4 put the boundary at 256 groups = 65 536 partial sums; the trace of a C2
iteration, profiles/r4_c2_kernel_durations.txt, shows that one block spending 22 us on 32 768 x 3 partial sums.)
In-process A/B of PSP_FOLD_ONE_BLOCK_GROUPS on the same buffers, Jacobi-PCG and Jacobi-MINRES, device-resident vectors,
best of three rounds; the iterates must be the same bits for every setting (both routes add in the canonical order)."""
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
    (4096, 4096, 0), (256, 256, 256), (2048, 2048, 0), (128, 128, 128), (1024, 1024, 0), (3000, 3000, 0), (512, 512, 512)]
settings = ("256", "64", "32", "16", "8", "4")
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
        iters = 100 if n >= (1 << 26) else 400
        for name, fn in (("pcg", L.psp_pcg_dev), ("minres", L.psp_minres_dev)):
            best = {s: 1e9 for s in settings}
            res = {}
            for rnd in range(4):
                for s in settings:
                    os.environ["PSP_FOLD_ONE_BLOCK_GROUPS"] = s
                    x.zero()
                    check(L.psp_synchronize())
                    t = time.perf_counter()
                    check(fn(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, iters, C.byref(info), C.byref(it), C.byref(rr), None))
                    check(L.psp_synchronize())
                    dt = time.perf_counter() - t
                    if rnd:
                        best[s] = min(best[s], dt)
                    if rnd == 3:
                        res[s] = (info.value, it.value, rr.value, float(np.abs(x.download()).sum()))
            row = {"partials": (n + 511) // 512, "groups": ((n + 511) // 512 + 255) // 256,
                   "us_per_iter": {s: round(best[s] / iters * 1e6, 2) for s in settings},
                   "same_bits": len(set(res.values())) == 1}
            print("x".join(str(g) for g in grid if g), form, name, json.dumps(row), flush=True)
        del A, K, aop, kop, b, x
