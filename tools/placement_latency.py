#!/usr/bin/env python3
"""Does the placement level of a vector show in its TRANSLATION / load latency?  (profiles/r4_modes.txt ruled the translation
reach out with a throughput probe -- one double per 4 KiB page, massively parallel; a dependent chain sees latency instead.)
One process: the 512^3 operator, x0, and eight candidates for y (pads in between); per candidate the SpMV time, then one
lane walking dependent loads through that buffer with strides of 256 B, 4 KiB + 64 B, 64 KiB + 64 B, 2 MiB + 64 B:
first pass (cold) and second pass (the lines of the first pass may sit in L2; the translations in the TLBs).  If a slow
buffer were mapped in smaller fragments, its 64 KiB / 2 MiB strides would cost a page walk per step where a fast buffer's
do not."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench  # noqa: E402
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

L = lib()
M = C.CDLL(os.path.join(HERE, "libmodeprobe.so"))
M.mp_chase_ns.restype = C.c_double
M.mp_chase_ns.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_int]
A = dev.DeviceCSR.poisson(512, 512, 512)
n = A.shape[0]
x0 = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
ev = bench.Events(L, check, 64)


def sync():
    check(L.psp_synchronize())


def spmv_ms(y):
    f = lambda: A.matvec_dev(x0.ptr, y.ptr)  # noqa: E731
    bench.timed_launches(f, sync, ev, 5)
    return round(bench.timed_launches(f, sync, ev, 30)[0], 4)


keep, cands = [], []
for j in range(8):
    keep.append(dev.DeviceBuffer((37 + 211 * j) * (1 << 17) + 512 * j))
    cands.append(dev.DeviceBuffer(n))
# a trivial kernel with the same number of streams (tools/modeprobe.hip mp_r7w1_k: seven read streams 1 GiB apart in one
# allocation of its own + one write stream = the candidate): does IT see the candidate's level?
M.mp_alloc.restype = C.c_void_p
M.mp_alloc.argtypes = [C.c_size_t]
M.mp_stream_ms.restype = C.c_double
M.mp_stream_ms.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int]
seven = M.mp_alloc(7 * n * 8) if os.environ.get("PLACE_R7W1") else None
strides = (("256B", 32), ("4KiB+64", 512 + 8), ("64KiB+64", 8192 + 8), ("2MiB+64", 262144 + 8))
for j, y in enumerate(cands):
    row = {"y": j, "addr": hex(y.ptr), "spmv_ms": spmv_ms(y)}
    sync()
    for name, st in strides:
        wrap = min(n // st, 20000)
        row[name + "_cold_ns"] = round(M.mp_chase_ns(y.ptr, n, st, 64, wrap), 1)
        row[name + "_2nd_ns"] = round(M.mp_chase_ns(y.ptr, n, st, wrap, wrap), 1)
    # the same launch with write-back instead of non-temporal stores of y / plain instead of non-temporal value loads
    DV = 194 + (64 << 8) + (1 << 20) + (1 << 22)
    for nm, f in (("plain_y_stores_ms", 2), ("plain_val_loads_ms", 1)):
        A.set_variant(DV + (f << 23))
        row[nm] = spmv_ms(y)
    A.set_variant(-1)
    row["r7w1_ms"] = round(M.mp_stream_ms(4, seven, y.ptr, n // 2, 8), 4) if seven else None
    row["spmv_ms_again"] = spmv_ms(y)
    print(json.dumps(row), flush=True)
