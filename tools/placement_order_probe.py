#!/usr/bin/env python3
"""Does the ORDER of a job's allocations decide the placement level of its first product?  One process per order:
  "operator-first": the 512^3 operator, then x, then y (what bench.py and most callers do);
  "vectors-first":  x, then y, then the operator.
Prints the average of 30 launches.  Run several fresh processes of each (tools/placement_order_probe.sh)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

L = lib()
order = sys.argv[1] if len(sys.argv) > 1 else "operator-first"
n = 512 ** 3
if order == "vectors-first":
    x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
    y = dev.DeviceBuffer(n)
    A = dev.DeviceCSR.poisson(512, 512, 512)
else:
    A = dev.DeviceCSR.poisson(512, 512, 512)
    x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
    y = dev.DeviceBuffer(n)
ev = bench.Events(L, check, 64)


def sync():
    check(L.psp_synchronize())


f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
bench.timed_launches(f, sync, ev, 5)
print(json.dumps({"order": order, "ms": round(bench.timed_launches(f, sync, ev, 30)[0], 4), "x": hex(x.ptr), "y": hex(y.ptr)}), flush=True)
