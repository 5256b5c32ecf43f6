#!/usr/bin/env python3
"""A/B the csr_spmv_stream variants on one GPU, interleaved rounds in ONE process
(cdna_hip_programming.md section 5.4 rule 24).  Prints one JSON line per variant."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402


def time_launches(fn, reps):
    L = lib()
    e0, e1 = C.c_void_p(), C.c_void_p()
    check(L.psp_event_create(C.byref(e0)))
    check(L.psp_event_create(C.byref(e1)))
    check(L.psp_event_record(e0))
    for _ in range(reps):
        fn()
    check(L.psp_event_record(e1))
    ms = C.c_float()
    check(L.psp_event_elapsed_ms(e0, e1, C.byref(ms)))
    L.psp_event_destroy(e0)
    L.psp_event_destroy(e1)
    return ms.value / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="512,512,512")
    ap.add_argument("--variants", default="4,20,32,36,48,52,40,44,60")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    nx, ny, nz = (int(t) for t in a.grid.split(","))
    print(dev.device_info(), flush=True)
    A = dev.DeviceCSR.poisson(nx, ny, nz)
    n, nnz = A.shape[0], A.nnz
    bytes_alg = 12 * nnz + 20 * n + 4
    x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
    y = dev.DeviceBuffer(n)
    variants = [int(v) for v in a.variants.split(",")]
    times = {v: [] for v in variants}
    # reference streaming rates on the same device: d2d copy and a dot (read-only)
    L = lib()
    big = dev.DeviceBuffer(n)
    t_copy = min(time_launches(lambda: check(L.psp_k_pupdate(n, x.ptr, None, 0.0, 1, big.ptr)), 10)
                 for _ in range(3))
    s = dev.DeviceBuffer(4)
    t_dot = min(time_launches(lambda: check(L.psp_k_dot(n, x.ptr, big.ptr, s.ptr)), 10) for _ in range(3))
    print(json.dumps({"copy_GBps": 16 * n / t_copy / 1e6, "dot_GBps": 16 * n / t_dot / 1e6}), flush=True)
    for v in variants:
        A.set_variant(v)
        time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 3)
    for _ in range(a.rounds):
        for v in variants:
            A.set_variant(v)
            times[v].append(time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), a.reps))
    for v in variants:
        t = np.array(times[v])
        print(json.dumps({"variant": v, "ms_med": float(np.median(t)), "ms_min": float(t.min()),
                          "GBps_med": bytes_alg / np.median(t) / 1e6, "frac_8TB": bytes_alg / np.median(t) / 1e6 / 8000}),
              flush=True)


if __name__ == "__main__":
    main()
