#!/usr/bin/env python3
"""Do the fused vector kernels lose time when their vectors sit at the SAME offset modulo a large power of two (separately
allocated vectors of n = 2^k doubles all do)?  px_update (3 reads + 2 writes over r, p, x) and r_update (2 reads + 1 write
over q, r) on vectors carved out of ONE allocation at spacing n*8 bytes exactly, against the same vectors staggered by
pads of (j * 4096 + j * 256) * 17 bytes; same process, alternated; includes the launches of the reduction behind each kernel."""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

L = lib()
for n in (1 << 24, 1 << 27, 12250000, 100000000):
    slack = 1 << 22
    big = dev.DeviceBuffer(5 * n + slack)
    check(L.psp_memset(big.ptr, 0, big.nbytes))
    out = dev.DeviceBuffer(8)
    res = {}
    for name, pad in (("aligned", 0), ("staggered", (4096 + 256) * 17), ("staggered_64k", 65536 + 4096 + 256)):
        ptr = [big.ptr + j * (8 * n + pad) for j in range(5)]
        r, p, x, q = ptr[0], ptr[1], ptr[2], ptr[3]
        fpx = lambda: check(L.psp_k_px_update(n, r, None, 0.5, 0, 0.25, 1, p, x, out.ptr))  # noqa: E731
        fr = lambda: check(L.psp_k_r_update(n, 0.25, q, None, r, out.ptr))  # noqa: E731
        res[name] = [1e9, 1e9]
        for f, k in ((fpx, 0), (fr, 1)):
            time_launches(f, 5)
    for rnd in range(4):
        for name, pad in (("aligned", 0), ("staggered", (4096 + 256) * 17), ("staggered_64k", 65536 + 4096 + 256)):
            ptr = [big.ptr + j * (8 * n + pad) for j in range(5)]
            r, p, x, q = ptr[0], ptr[1], ptr[2], ptr[3]
            fpx = lambda: check(L.psp_k_px_update(n, r, None, 0.5, 0, 0.25, 1, p, x, out.ptr))  # noqa: E731
            fr = lambda: check(L.psp_k_r_update(n, 0.25, q, None, r, out.ptr))  # noqa: E731
            res[name][0] = min(res[name][0], time_launches(fpx, 20))
            res[name][1] = min(res[name][1], time_launches(fr, 20))
    print(json.dumps({"n": n, "log2n": round(float(np.log2(n)), 2),
                      **{k: {"px_update_ms": round(v[0], 4), "r_update_ms": round(v[1], 4),
                             "px_TBps": round(40 * n / v[0] / 1e9, 3), "r_TBps": round(24 * n / v[1] / 1e9, 3)} for k, v in res.items()}}),
          flush=True)
    big.free()
    out.free()
