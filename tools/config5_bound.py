#!/usr/bin/env python3
"""tools/config5_bound.py -- what bounds the irregular sss_mat product of BASELINE.json configs[4] (round 6, VERDICT r5 #4b).

SSS-model bytes (12 nnz_lower + 28 n + 4) price a product that reads the strict lower triangle ONCE.  The
product that runs multiplies with the expanded mirror (every off-diagonal value and 16-bit column twice) because the
reference's summation order per row (sss_mat.c:45-55) is then a plain row sum -- no atomics, the oracle's bits.  This
script measures, at the stand-in's size and in ONE launch each (ramp included, like the product itself):
  * plain streaming kernels (psp_stream_probe: 7 reads + 1 write, the product's shape) over the bytes of the SSS model and
    over the bytes the mirror's format has to move -- the ceilings of a lower-triangle-only product and of the mirror product;
  * the product through the API (with its two permutation passes) and the Jacobi-MINRES iteration (whose product runs in the
    copy's numbering without them).
Output: one JSON object; profiles/r6_config5_bound.txt quotes it beside the counters of tools/pmc_fem.sh."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysparse_amd import device as dev  # noqa: E402
from pysparse_amd._capi import check, lib  # noqa: E402
from pysparse_amd.tools import standins  # noqa: E402


def probe(L, total_bytes, reads=7, writes=1, reps=50):
    per = int(total_bytes / (reads + writes)) // 4096 * 4096
    avg, mn = C.c_float(), C.c_float()
    check(L.psp_stream_probe(reads, writes, C.c_size_t(per), reps, C.byref(avg), C.byref(mn)))
    b = per * (reads + writes)
    return {"bytes": b, "avg_us": avg.value * 1e3, "min_us": mn.value * 1e3, "GBps_avg": b / (avg.value * 1e-3) / 1e9}


def main():
    shuffle = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    L = lib()
    n, ind, col, val, diag = standins.fem_sss_arrays(68, 68, 67, shuffle)
    nl = int(val.shape[0])
    nnz = 2 * nl + n
    sss_bytes = 12 * nl + 28 * n + 4
    chunks = nnz / 1016.0
    mirror_bytes = int(10 * nnz * (1024 / 1016.0) + chunks * (4 * 64 + 2 * 256 + 16) + 16 * n)  # bench_common.kernel_bytes (csr_spmv_w3)
    out = {"standin": "fem%d" % shuffle, "n": n, "nnz_lower": nl, "nnz_full": nnz, "sss_model_bytes": sss_bytes,
           "mirror_format_bytes": mirror_bytes, "mirror_over_sss": mirror_bytes / sss_bytes}
    out["stream_sss_bytes"] = probe(L, sss_bytes)
    out["stream_mirror_bytes"] = probe(L, mirror_bytes)
    out["stream_mirror_bytes_read_only"] = probe(L, mirror_bytes, 8, 0)
    out["stream_8GB"] = probe(L, 8 << 30, reps=10)
    S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
    S.prepare(1 << 30)
    x = dev.DeviceBuffer.from_host(np.random.default_rng(1).standard_normal(n))
    y = dev.DeviceBuffer(n)
    S.matvec_dev(x.ptr, y.ptr)
    check(L.psp_synchronize())
    out["kernel"] = S.kernel_info()[0]
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(100):
            S.matvec_dev(x.ptr, y.ptr)
        check(L.psp_synchronize())
        ts.append((time.perf_counter() - t0) / 100)
    t = min(ts)
    out["product_api_us"] = t * 1e6
    out["product_api_frac_sss_model"] = sss_bytes / t / 8e12
    K = dev.DeviceJacobi(S)
    b = np.zeros(n)
    b[0] = 1.0
    times = {}
    for k in (20, 220):
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            dev.minres(S, b, np.zeros(n), 0.0, k, K)
            best = min(best, time.perf_counter() - t0)
        times[k] = best
    out["minres_us_per_iteration"] = (times[220] - times[20]) / 200 * 1e6
    # the ceilings, as fractions of the 8 TB/s peak in SSS-model bytes
    out["ceiling_lower_only_frac"] = sss_bytes / (out["stream_sss_bytes"]["avg_us"] * 1e-6) / 8e12
    out["ceiling_mirror_frac"] = sss_bytes / (out["stream_mirror_bytes"]["avg_us"] * 1e-6) / 8e12
    print(json.dumps(out))


if __name__ == "__main__":
    main()
