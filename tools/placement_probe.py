#!/usr/bin/env python3
"""Which operand's placement sets the launch time of csr_spmv_w4 at 512^3?  bench.py's placement sweep saw six fresh y
allocations (made after the w3 / w2 layouts, 12 GB, had been built) all at 1.497 ms while the job's first y ran at
1.618 ms.  Here, in one process, with the addresses printed:

  1. (x0, y0) as a job allocates them          2. y1..y3 fresh, pads in between        3. y0 again
  4. after allocating a 12 GB block: y4..y6    5. y0 again, x1 fresh with y0, x1 with y4

Each timing is the average of 30 back-to-back launches after 5 warm-ups, HIP events on the library's stream."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402

L = lib()
grid = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "512,512,512").split(","))
big_gb = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
A = dev.DeviceCSR.poisson(*grid)
n = A.shape[0]
x0 = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y0 = dev.DeviceBuffer(n)
ev = bench.Events(L, check, 64)


def sync():
    check(L.psp_synchronize())


def t(x, y):
    f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
    bench.timed_launches(f, sync, ev, 5)
    return round(bench.timed_launches(f, sync, ev, 30)[0], 4)


ob = dev.DeviceBuffer(64)


def tk(fn):
    bench.timed_launches(fn, sync, ev, 5)
    return round(bench.timed_launches(fn, sync, ev, 30)[0], 4)


rows = []


def rec(label, x, y):
    rows.append({"case": label, "x": hex(x.ptr), "y": hex(y.ptr), "y_minus_x_MiB": round((y.ptr - x.ptr) / 2 ** 20, 3), "ms": t(x, y),
                 # the same pair under two plain streaming kernels of the library: z = d * r (2 reads of x, 1 write of y), then x . y (2 reads)
                 "triad_write_y_ms": tk(lambda: check(L.psp_k_jacobi(n, x.ptr, x.ptr, y.ptr))),
                 "dot_read_xy_ms": tk(lambda: check(L.psp_k_dot(n, x.ptr, y.ptr, ob.ptr))),
                 "dot_read_yy_ms": tk(lambda: check(L.psp_k_dot(n, y.ptr, y.ptr, ob.ptr))),
                 "dot_read_xx_ms": tk(lambda: check(L.psp_k_dot(n, x.ptr, x.ptr, ob.ptr)))})
    print(json.dumps(rows[-1]), flush=True)


keep = []
rec("x0,y0 first", x0, y0)
for j in range(3):
    keep.append(dev.DeviceBuffer((37 + 101 * j) * (1 << 17) + 512 * j))
    y = dev.DeviceBuffer(n)
    keep.append(y)
    rec("x0,y%d fresh after pad" % (j + 1), x0, y)
rec("x0,y0 again", x0, y0)
big = dev.DeviceBuffer(int(big_gb * 2 ** 30 / 8))
keep.append(big)
ys = []
for j in range(3):
    keep.append(dev.DeviceBuffer((37 + 101 * j) * (1 << 17) + 512 * j))
    y = dev.DeviceBuffer(n)
    ys.append(y)
    rec("x0,y%d fresh after the %g GB block" % (j + 4, big_gb), x0, y)
rec("x0,y0 again", x0, y0)
x1 = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
rec("x1 fresh,y0", x1, y0)
rec("x1 fresh,y4", x1, ys[0])
rec("x0,y0 last", x0, y0)
