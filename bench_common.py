"""bench_common.py -- what bench.py, bench_launch.py and bench_legs.py share: the metric's constants, the byte models
(DESIGN.md section 3), HIP-event timing on the library's stream, provenance.  No GPU is touched at
import time and nothing here imports the oracle."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3-6.8 achievable)


W3_VARIANT = (1 << 20) + 128 + 64 + 2 + (64 << 8)  # csr_spmv_w3 (general banded CSR), see psp_csr.hip


W2_VARIANT = 128 + 64 + 2 + (64 << 8)              # csr_spmv_w2 (int32 col + fp64 val streamed as stored)
W6_VARIANT = W2_VARIANT + (1 << 23)                # csr_spmv_w6 (the same streams, x staged in LDS; round 5)


PMC_FILES = {"csr_spmv_w4": "r6_spmv_pmc.json", "csr_spmv_w3": "r5_spmv_w3_pmc.json",
             "csr_spmv_w6": "r6_spmv_w6_pmc.json", "csr_spmv_w2": "r5_spmv_w2_pmc.json"}


def csr_model_bytes(n, nnz):
    """val 8 + col 4 per nonzero; ind 4 + y 8 + x 8 per row."""
    return 12 * nnz + 20 * n + 4


def kernel_bytes(kernel, info, n, nnz, nnz_lower=None):
    """Bytes one launch of `kernel` has to move from/to DRAM: x and y once + the matrix in the
    format that kernel streams.  Never more than the CSR model."""
    if kernel == "csr_spmv_w4":   # values in padded offset-major blocks of 128 rows + 16-bit row masks
        rows = (n + 127) // 128 * 128
        return 8 * rows * info["nb"] + 2 * n + 16 * n
    if kernel == "sss_spmv_w4":   # strict lower triangle (offset-major), diagonal, 16-bit masks
        rows = (n + 127) // 128 * 128
        return 8 * rows * info["nb"] + 8 * n + 2 * n + 16 * n
    if kernel == "csr_spmv_w3":   # val 8 + col16 2 per nonzero; per chunk of ~1016 nonzeros: block list + row offsets
        chunks = nnz / 1016.0
        return int(10 * nnz * (1024 / 1016.0) + chunks * (4 * info["nb"] + 2 * 256 + 16) + 16 * n)
    return csr_model_bytes(n, nnz)


def pcg_vector_bytes(n, lazy, const_dinv=True):
    """vector traffic of one fused Jacobi-PCG iteration beside the SpMV:
    lazy: px_update (r, p, x read; p, x written) + r_update (q, r read; r written) = 64 n;
    eager: pupdate 24 n + x_update 24 n + r_update 24 n; + 16 n when dinv is streamed (twice)."""
    return (64 if lazy else 72) * n + (0 if const_dinv else 16 * n)


class Events:
    """HIP events on the library's stream (the stream the kernels are launched on)."""

    def __init__(self, L, check, count=2):
        self.L, self.check = L, check
        self.ev = []
        for _ in range(count):
            e = C.c_void_p()
            check(L.psp_event_create(C.byref(e)))
            self.ev.append(e)

    def record(self, i):
        self.check(self.L.psp_event_record(self.ev[i]))

    def ms(self, i, j):
        ms = C.c_float()
        self.check(self.L.psp_event_elapsed_ms(self.ev[i], self.ev[j], C.byref(ms)))
        return float(ms.value)


def timed_launches(step, sync, ev, count):
    """count launches, one event between each: (average ms, median ms) per launch"""
    sync()
    for i in range(count):
        ev.record(i)
        step()
    ev.record(count)
    sync()
    per = [ev.ms(i, i + 1) for i in range(count)]
    return ev.ms(0, count) / count, float(np.median(per))


METRIC = "CSR SpMV GB/s (7-pt Poisson, % of 8 TB/s HBM peak) + PCG iters/s"


PARITY_ITERS = 20      # iterations of the in-job parity solves (tol = 0)


PARITY_TOL = 1e-9      # N-rank solve against the one-GPU solve of the same problem: relres and x checksums


def rel_diff(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)


def parity_object(n1, nr, what):
    """`parity_vs_n1`: the N-rank Jacobi-PCG against the one-GPU solve of the same system after PARITY_ITERS
    iterations (tol = 0): the recurred residual and two checksums of x.  The two differ by the order of the
    reductions only (<= 1e-13 on the probes)."""
    d = {k: rel_diff(n1[k], nr[k]) for k in ("relres", "x_dot_b", "x_dot_x")}
    worst = max(d.values())
    return {"iters": PARITY_ITERS, "n1": n1, what: nr, "rel_diff": d, "max_rel_diff": worst, "tol": PARITY_TOL,
            "same_info_iter": n1["info_iter"] == nr["info_iter"], "ok": bool(worst <= PARITY_TOL and
                                                                               n1["info_iter"] == nr["info_iter"])}


def provenance(L):
    """which sources the library that ran was built from (tests/test_capi_symbols.py holds the two equal)"""
    out = {"build_id": L.psp_build_id().decode()}
    try:
        import __graft_entry__ as G
        out["source_hash"] = G.source_hash()
        out["match"] = out["build_id"] == out["source_hash"]
    except Exception as e:  # noqa: BLE001 - the sources may not lie next to an installed library
        out["source_hash"] = None
        out["match"] = None
        out["note"] = str(e)[:120]
    return out


def _maxrel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
