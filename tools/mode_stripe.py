#!/usr/bin/env python3
"""Timing modes vs the XCD-stripe remap: in ONE fresh process, csr_spmv_w4 at 512^3 with stripe 32 (default), 0 (plain
dispatch order), 8 and 128, interleaved rounds.  If the process-to-process spread is the dispatcher's order falling differently
against the remap, the stripe-0 time should not move with the default's."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

BASE = 128 + 2 + 64 + (1 << 20) + (1 << 22)  # w1 | w2 | full grid | w3 bit | w4 bit (psp_csr.hip kDefaultVariant)
if os.environ.get("KERNEL") == "w3":
    BASE -= 1 << 22
if os.environ.get("KERNEL") == "w2":
    BASE -= (1 << 22) + (1 << 20)
GRID = tuple(int(t) for t in os.environ.get("GRID", "512,512,512").split(","))
if os.environ.get("SSS") == "1":
    A = dev.DeviceSSS.poisson(*GRID)
elif os.environ.get("BIG") == "1":
    A = dev.DeviceCSR.poisson_big(*GRID)
else:
    A = dev.DeviceCSR.poisson(*GRID)
n = A.shape[0]
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
out = {}
A.set_variant(-1)
time_launches(f, 10)
DEFAULT = -1 if not os.environ.get("KERNEL") else BASE + (64 << 8)
for rnd in range(4):
    stripes = [int(t) for t in os.environ.get("STRIPES", "0,8,32,128").split(",")]
    for name, var in [("default", DEFAULT)] + [("stripe%d" % t, BASE + (t << 8)) for t in stripes]:
        A.set_variant(var)
        time_launches(f, 3)
        out.setdefault(name, []).append(round(time_launches(f, 20), 4))
A.set_variant(-1)
A.set_variant(DEFAULT)
print(json.dumps({"grid": GRID, "sss": os.environ.get("SSS") == "1", "kernel": A.kernel_info()[0]}), json.dumps({k: min(v) for k, v in out.items()}))
