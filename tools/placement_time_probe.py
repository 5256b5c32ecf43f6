#!/usr/bin/env python3
"""tools/placement_time_probe.py -- is the "placement level" of the 512^3 product a property of WHERE y lies or of WHEN it is
measured (round 6)?  One process: the first (x, y) pair is timed, the GPU is kept busy, the SAME pair is timed again, then new
y vectors are allocated and timed alternately with the old one."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from bench_common import Events, timed_launches  # noqa: E402

L = lib()


def sync():
    check(L.psp_synchronize())


A = dev.DeviceCSR.poisson(512, 512, 512)
n = A.shape[0]
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y0 = dev.DeviceBuffer(n)
ev = Events(L, check, 60)
t_start = time.perf_counter()


def t(y, k=20):
    f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
    timed_launches(f, sync, ev, 3)
    return round(timed_launches(f, sync, ev, k)[0], 4)


out = {"y0_first": t(y0)}
for burst in (200, 1000, 2000):
    for _ in range(burst):
        A.matvec_dev(x.ptr, y0.ptr)
    sync()
    out["y0_after_%d_more_launches" % burst] = t(y0)
    out["t_%d_s" % burst] = round(time.perf_counter() - t_start, 2)
ys = [y0]
seq = []
for j in range(6):
    pad = dev.DeviceBuffer((37 + 101 * j) * (1 << 17) + 512 * j)
    ys.append(pad)
    yj = dev.DeviceBuffer(n)
    ys.append(yj)
    seq.append({"new_y": t(yj), "y0_again": t(y0)})
out["sweep"] = seq
print(json.dumps(out))
