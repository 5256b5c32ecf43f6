#!/usr/bin/env python3
"""A/B the plane-sweeping workgroup schedule of csr_spmv_w3 (psp_csr_set_schedule) in ONE
process, interleaved rounds.  strip_rows 0 = natural order + XCD stripes."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from tools.spmv_sweep import time_launches  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="512,512,512")
    ap.add_argument("--strips", default="0,2048,4096,8192,16384,32768")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    nx, ny, nz = (int(t) for t in a.grid.split(","))
    A = dev.DeviceCSR.poisson(nx, ny, nz)
    A.set_variant(a.variant)
    n, nnz = A.shape[0], A.nnz
    bytes_alg = 12 * nnz + 20 * n + 4
    x = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n))
    y = dev.DeviceBuffer(n)
    strips = [int(s) for s in a.strips.split(",")]
    times = {s: [] for s in strips}
    yref = None
    for s in strips:
        A.set_schedule(s)
        y.zero()  # a launch that does nothing must not pass on the previous result
        time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 2)
        yh = y.download()
        if yref is None:
            yref = yh
        assert np.array_equal(yh, yref), "schedule changed the result"
        print(json.dumps({"strip_rows": s, "kernel": A.kernel_info()}), flush=True)
    for _ in range(a.rounds):
        for s in strips:
            A.set_schedule(s)
            time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), 1)
            times[s].append(time_launches(lambda: A.matvec_dev(x.ptr, y.ptr), a.reps))
    for s in strips:
        t = np.array(times[s])
        print(json.dumps({"strip_rows": s, "ms_med": float(np.median(t)), "ms_min": float(t.min()),
                          "GBps_med": bytes_alg / np.median(t) / 1e6,
                          "frac_8TB": bytes_alg / np.median(t) / 1e6 / 8000}), flush=True)


if __name__ == "__main__":
    main()
