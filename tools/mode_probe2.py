#!/usr/bin/env python3
"""One process, several probes: which of them moves together with the csr_spmv_w4 time of this process?
Prints one JSON line: the w4 time at 512^3, streaming probes through tools/libmodeprobe.so (read / fill / copy /
7-reads-1-write, one 16-byte element per thread, 1 GiB per stream) and a census of the hardware ids the waves of a
262144-workgroup grid report (XCC_ID, and the CU / SE / VMID / queue / pipe / ME fields of HW_ID)."""
import collections
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402

M = C.CDLL(os.path.join(HERE, "libmodeprobe.so"))
M.mp_alloc.restype = C.c_void_p
M.mp_alloc.argtypes = [C.c_size_t]
M.mp_free.argtypes = [C.c_void_p]
M.mp_stream_ms.restype = C.c_double
M.mp_stream_ms.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_int]
M.mp_census.argtypes = [C.c_int, C.c_void_p, C.c_void_p]

n = 512 ** 3
A = dev.DeviceCSR.poisson(512, 512, 512)
x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
y = dev.DeviceBuffer(n)
f = lambda: A.matvec_dev(x.ptr, y.ptr)  # noqa: E731
time_launches(f, 10)
w4 = min(time_launches(f, 20) for _ in range(5))

GiB = 1 << 30
n2 = GiB // 16
a = M.mp_alloc(7 * GiB)
b = M.mp_alloc(GiB)
out = {"w4_ms": round(w4, 4)}
for kind, name, nbytes in ((0, "read", GiB), (1, "fill", GiB), (3, "fill_nt", GiB), (2, "copy", 2 * GiB),
                           (4, "r7w1", 8 * GiB)):
    ms = M.mp_stream_ms(kind, a, b, n2, 5)
    out[name + "_GBps"] = round(nbytes / ms / 1e6, 1)
w4b = min(time_launches(f, 20) for _ in range(3))
out["w4_ms_after"] = round(w4b, 4)
nb = 262144
hw = np.zeros(nb, dtype=np.uint32)
xc = np.zeros(nb, dtype=np.uint32)
M.mp_census(nb, hw.ctypes.data, xc.ctypes.data)
xcc = xc & 0xF
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
fields = {"vmid": (hw >> 20) & 0xF, "queue": (hw >> 24) & 0x7, "pipe": (hw >> 6) & 0x3, "me": (hw >> 30) & 0x3}
for k, v in fields.items():
    out[k] = sorted(collections.Counter(v.tolist()).items())
where = xcc.astype(np.int64) * 4096 + se * 256 + sh * 16 + cu
cnt = collections.Counter(where.tolist())
out["distinct_cu"] = len(cnt)
out["wg_per_cu_min_max"] = [min(cnt.values()), max(cnt.values())]
out["wg_per_xcc"] = [int((xcc == i).sum()) for i in range(8)]
out["x_ptr"], out["y_ptr"] = hex(x.ptr), hex(y.ptr)
print(json.dumps(out))
