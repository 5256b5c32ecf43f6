#!/usr/bin/env python3
"""Where the time between the kernels of a solver iteration goes: reads a `rocprofv3 --kernel-trace --output-format csv`
trace (the *_kernel_trace.csv under the given directory) and prints, per pair (kernel, next kernel) in stream order, the
number of boundaries and the mean / median gap  start(next) - end(kernel)  in microseconds, plus each kernel's own mean
duration.  Usage: python tools/gap_trace.py DIR [min_count]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

import numpy as np


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    name = re.sub(r"[<(].*", "", name)
    return name.split("::")[-1].strip()


def main():
    d = sys.argv[1]
    min_count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    gaps, durs = defaultdict(list), defaultdict(list)
    for (s0, e0, k0), (s1, e1, k1) in zip(rows[:-1], rows[1:]):
        durs[k0].append((e0 - s0) / 1e3)
        g = (s1 - e0) / 1e3
        if g < 200.0:  # a host round trip in between is not a launch gap
            gaps[(k0, k1)].append(g)
    print("# kernel durations (us): count mean median")
    for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= min_count:
            print("%-34s %7d %10.2f %10.2f" % (k, len(v), np.mean(v), np.median(v)))
    # busy fraction of the stretches in which kernels follow each other without a host round trip (gap < 200 us)
    seg_busy, seg_span, seg_n, cur0, cur_busy, n_in = 0.0, 0.0, 0, None, 0.0, 0
    longest = []
    for i, (s0, e0, k0) in enumerate(rows):
        if cur0 is None:
            cur0, cur_busy, n_in = s0, 0.0, 0
        cur_busy += e0 - s0
        n_in += 1
        last = i == len(rows) - 1 or (rows[i + 1][0] - e0) / 1e3 >= 200.0
        if last:
            if n_in >= 50:
                seg_busy += cur_busy
                seg_span += e0 - cur0
                seg_n += 1
                longest.append(((e0 - cur0) / 1e6, cur_busy / (e0 - cur0), n_in))
            cur0 = None
    if seg_span:
        print("# stretches of >= 50 back-to-back kernels: %d, kernel-busy fraction of their span %.4f" % (seg_n, seg_busy / seg_span))
        for ms, frac, cnt in sorted(longest, reverse=True)[:6]:
            print("#   stretch of %.2f ms, %d kernels, busy %.4f" % (ms, cnt, frac))
    print("# gaps between consecutive kernels (us): count mean median   [end of first -> start of second]")
    for (k0, k1), v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
        if len(v) >= min_count:
            print("%-34s -> %-34s %7d %8.2f %8.2f" % (k0, k1, len(v), np.mean(v), np.median(v)))


if __name__ == "__main__":
    main()
