#!/usr/bin/env python3
"""VERDICT r4 'Next' #5's measure for cgs / bicgstab / qmrs / gmres(20): the time of an iteration against the sum of the
durations of the kernels in it.

  kry_floor.py run SOLVER GRID ITERS        one truncated solve (tol = 0) -- the program to put under rocprofv3 --kernel-trace
  kry_floor.py time SOLVER GRID             microseconds per iteration from two truncated solves, no profiler (one JSON line)
  kry_floor.py sum DIR ITERS US_PER_ITER    reads DIR/**/*kernel_trace.csv: the kernels launched at least ITERS / 2 times are
                                            the iteration's; sum of (mean duration x launches) / ITERS against US_PER_ITER"""
import csv
import glob
import json
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def solve(solver, grid, iters_list):
    from pysparse_amd import device as dev
    A = dev.DeviceCSR.poisson(*grid)
    n = A.shape[0]
    K = dev.DeviceJacobi(A)
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    fn = getattr(dev, solver)
    ts = {}
    for k in iters_list:
        x = np.zeros(n)
        t = time.perf_counter()
        res = fn(A, b, x, 0.0, k, K)
        ts.setdefault(k, []).append(time.perf_counter() - t)
        assert res[1] in (k, k + 1), res
    return ts


def main():
    mode = sys.argv[1]
    if mode in ("run", "time"):
        solver = sys.argv[2]
        grid = tuple(int(t) for t in sys.argv[3].split(","))
        if mode == "run":
            solve(solver, grid, [int(sys.argv[4])])
            return
        k1, k2 = (100, 1100) if solver != "gmres" else (100, 500)
        ts = solve(solver, grid, [k1, k1, k2, k1, k2, k1, k2])
        print(json.dumps({"solver": solver, "grid": list(grid), "us_per_iter": (min(ts[k2]) - min(ts[k1])) / (k2 - k1) * 1e6}))
        return
    d, iters, us = sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    durs = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot, nlaunch = 0.0, 0
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= iters / 2:
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
            print("  %-72s %7.2f per iteration, mean %7.2f us" % (short, len(v) / iters, np.mean(v)))
            tot += sum(v)
            nlaunch += len(v)
    print("  kernels per iteration %.1f, their durations %.1f us, iteration %.1f us: ratio %.2f" % (
        nlaunch / iters, tot / iters, us, us / (tot / iters)))


if __name__ == "__main__":
    main()
