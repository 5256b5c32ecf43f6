#!/usr/bin/env python3
"""sss_spmv_w4: the lower triangle's values in 128-row blocks (offset-major inside a block: the product's first touch of a
block comes in three visits of 1 KiB, one per offset, far apart in time) against one array per offset (every stream of
the product contiguous).  Three handles of each layout (placement noise), the same x and y, interleaved, same bits."""
import json
import os
import sys

import numpy as np

os.environ["PSP_TUNING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

grid = tuple(int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "512,512,512").split(","))
copies = int(sys.argv[2]) if len(sys.argv) > 2 else 3
hs = []
for c in range(copies):
    for soa in ("0", "1"):
        os.environ["PSP_SSS_SOA"] = soa
        S = dev.DeviceSSS.poisson(*grid)
        hs.append((soa, S))
n = hs[0][1].n
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
ref = None
for soa, S in hs:
    os.environ["PSP_SSS_SOA"] = soa
    S.matvec_dev(x.ptr, y.ptr)  # builds the tables under the handle's own setting
for soa, S in hs:
    os.environ["PSP_SSS_SOA"] = soa
    y.zero()
    time_launches(lambda: S.matvec_dev(x.ptr, y.ptr), 2)
    yh = y.download()
    if ref is None:
        ref = yh
    assert np.array_equal(yh, ref), soa
times = [[] for _ in hs]
for _ in range(5):
    for i, (soa, S) in enumerate(hs):
        time_launches(lambda: S.matvec_dev(x.ptr, y.ptr), 1)
        times[i].append(time_launches(lambda: S.matvec_dev(x.ptr, y.ptr), 10))
for i, (soa, S) in enumerate(hs):
    print(json.dumps({"layout": "per-offset arrays" if soa == "1" else "128-row blocks", "kernel": S.kernel_info()[0],
                      "ms": round(float(np.median(times[i])), 4)}), flush=True)
