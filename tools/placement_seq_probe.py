#!/usr/bin/env python3
"""tools/placement_seq_probe.py SEQ -- round 6: does the ORDER in which the two vector operands are allocated decide the level of the
512^3 product?  SEQ is a string of steps: x, y (allocate the operand), d (a spacer of the same size, zeroed, kept), D (a spacer that
is freed again at once), k (20 products on what exists so far, when both exist)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from bench_common import Events, timed_launches  # noqa: E402

L = lib()
seq = sys.argv[1]


def sync():
    check(L.psp_synchronize())


A = dev.DeviceCSR.poisson(512, 512, 512)
n = A.shape[0]
A.kernel_info()
ev = Events(L, check, 60)
x = y = None
keep = []
for c in seq:
    if c == "x":
        x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
    elif c == "y":
        y = dev.DeviceBuffer(n)
    elif c == "d":
        b = dev.DeviceBuffer(n)
        b.zero()
        keep.append(b)
    elif c == "D":
        b = dev.DeviceBuffer(n)
        b.zero()
        sync()
        b.free()
    elif c == "k" and x is not None and y is not None:
        for _ in range(20):
            A.matvec_dev(x.ptr, y.ptr)
        sync()
f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
timed_launches(f, sync, ev, 5)
ms = timed_launches(f, sync, ev, 30)[0]
print(json.dumps({"seq": seq, "ms": round(ms, 4), "frac": round(9932111872 / (ms * 1e-3) / 8e12, 4)}))
