#!/usr/bin/env python3
"""What does the host <-> device path of this box give (tuning aid for the host-pointer matvec, csr_mat.c:141-163)?
1 GiB transfers: pageable / pinned, each way, both ways at once (two threads, two streams), hipHostRegister cost."""
import json
import threading
import time

import numpy as np
import torch

GiB = 1 << 30
n = GiB // 8
dev = torch.device("cuda", 0)
d1 = torch.empty(n, dtype=torch.float64, device=dev)
d2 = torch.empty(n, dtype=torch.float64, device=dev)
out = {}


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best


hp = torch.from_numpy(np.random.default_rng(0).standard_normal(n))
hq = torch.empty(n, dtype=torch.float64)
out["pageable_h2d_GBps"] = GiB / timeit(lambda: d1.copy_(hp)) / 1e9
out["pageable_d2h_GBps"] = GiB / timeit(lambda: hq.copy_(d2)) / 1e9
pp = torch.empty(n, dtype=torch.float64).pin_memory()
pq = torch.empty(n, dtype=torch.float64).pin_memory()
pp.copy_(hp)
out["pinned_h2d_GBps"] = GiB / timeit(lambda: d1.copy_(pp, non_blocking=True)) / 1e9
out["pinned_d2h_GBps"] = GiB / timeit(lambda: pq.copy_(d2, non_blocking=True)) / 1e9
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def both_pinned():
    with torch.cuda.stream(s1):
        d1.copy_(pp, non_blocking=True)
    with torch.cuda.stream(s2):
        pq.copy_(d2, non_blocking=True)


out["pinned_duplex_GBps_sum"] = 2 * GiB / timeit(both_pinned) / 1e9


def both_pageable():
    def up():
        with torch.cuda.stream(s1):
            d1.copy_(hp)
    t = threading.Thread(target=up)
    t.start()
    with torch.cuda.stream(s2):
        hq.copy_(d2)
    t.join()


out["pageable_duplex_two_threads_GBps_sum"] = 2 * GiB / timeit(both_pageable) / 1e9
# host memcpy rate (what a staging copy costs), 1 and 4 threads
a = np.empty(n)
src = hp.numpy()
t = time.perf_counter()
np.copyto(a, src)
out["host_memcpy_1thread_GBps"] = GiB / (time.perf_counter() - t) / 1e9


def part(i, k):
    lo, hi = i * n // k, (i + 1) * n // k
    np.copyto(a[lo:hi], src[lo:hi])


for k in (4, 8):
    ths = [threading.Thread(target=part, args=(i, k)) for i in range(k)]
    t = time.perf_counter()
    [th.start() for th in ths]
    [th.join() for th in ths]
    out["host_memcpy_%dthreads_GBps" % k] = GiB / (time.perf_counter() - t) / 1e9
# registration cost
rt = torch.cuda.cudart()
buf = np.empty(n)
buf[:] = 1.0
t = time.perf_counter()
rc = rt.cudaHostRegister(buf.ctypes.data, GiB, 0)
out["hostRegister_1GiB_ms"] = (time.perf_counter() - t) * 1e3
out["hostRegister_rc"] = int(rc)
if int(rc) == 0:
    tb = torch.from_numpy(buf)
    out["registered_h2d_GBps"] = GiB / timeit(lambda: d1.copy_(tb, non_blocking=True)) / 1e9
    t = time.perf_counter()
    rt.cudaHostUnregister(buf.ctypes.data)
    out["hostUnregister_ms"] = (time.perf_counter() - t) * 1e3
print(json.dumps(out, indent=1))
