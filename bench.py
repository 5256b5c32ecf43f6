#!/usr/bin/env python3
"""bench.py -- CSR SpMV GB/s (% of HBM peak) + Jacobi-PCG iterations/s, 7-pt Poisson, fp64.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]

N > 1 without a torch.distributed environment: bench.py starts its own ranks (a fresh
`python -m torch.distributed.run --nproc-per-node N` child, before anything touches the GPU)
and relays rank 0's JSON line; started under torch.distributed.run it is one rank of N.

A "step" is one y = A x with the operator resident in HBM (N > 1: one ghost exchange of x over
RCCL + the local SpMV of every rank).  W untimed warm-up steps, then exactly K steps between
barrier + synchronize on both sides; MAX over ranks; rank 0 prints ONE JSON line.

  N = 1 (default)      512^3 grid (BASELINE.json configs[2]); beside it the 1024^3 operator on the
                       same GPU (PCG iterations/s: the 1-GPU end of the strong-scaling target)
  N > 1 (default)      --scaling strong: the fixed 1024^3 grid (configs[3]) cut into z-slabs of
                       1024/N planes, index-free slab operator (per-rank nnz exceeds 32 bits at
                       N = 2); rank 0 first times the whole 1024^3 problem alone (`strong_n1`)
  --scaling weak       1024 x 1024 x 128 N grid, 2^27 rows per GPU (N = 8: the same 1024^3 problem)
  --gpus 1 --scaling strong   the 1024^3 problem through the multi-GPU driver at world size 1

`value` = bytes the SpMV kernel that ran HAS TO MOVE (its own matrix format + x once + y once,
DESIGN.md section 3) / time, so no fraction of the 8 TB/s peak can exceed 1; the rate in CSR-model
bytes (12 nnz + 20 n + 4) is printed beside it as `effective_csr_model_GBps`.
"""
import os

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")  # cpu_baseline legs: ONE core, also inside OpenBLAS

import argparse  # noqa: E402
import ctypes as C  # noqa: E402
import json  # noqa: E402
import re  # noqa: E402
import socket  # noqa: E402
import subprocess  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402

from bench_common import (HBM_PEAK_GBPS, METRIC, PARITY_ITERS, PARITY_TOL, PMC_FILES, ROOT, W2_VARIANT, W3_VARIANT,  # noqa: E402,F401
                          Events, _maxrel, csr_model_bytes, kernel_bytes, parity_object, pcg_vector_bytes, provenance,
                          rel_diff, timed_launches)
from bench_launch import LADDER, guarded_rank, orchestrate  # noqa: E402
from bench_line import emit, judge_phases, predicted_iteration  # noqa: E402
from bench_legs import (dry_strong_n1, gpu_clocks, link_topology, live_traffic, pcg_single, peer_matrix,  # noqa: E402
                        placement_sweep_leg, same_operator_kernels_leg, single_kernel_leg, single_process_main, solvers_leg, sss_leg, stream_ceiling_leg,
                        strong_n1_leg)













# ------------------------------------------------------------------------------------ CPU legs

def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


REF_PCG_TOL = 1e-9  # max-norm relative difference allowed between the reference's and the port's PCG iterates


def parity_bound(n, k, against="oracle"):
    """How far two correct implementations of the same k Krylov iterations may be apart at n unknowns when they differ
    only in the ORDER of their dot-product sums (sequential loop, OpenBLAS kernels, the GPU's fixed tree): every
    reduction of n terms carries ~sqrt(n) eps of order-dependent rounding, each iteration passes it on through alpha /
    beta, so the iterates drift apart like k sqrt(n) eps.  The head-room over that depends on who is compared:
      against="oracle"     32 x: the oracle adds its n terms one after the other (the worst order there is); it is itself
                           4.9e-11 from the compiled reference at 512^3 / k = 3, where this gives 2.5e-10;
      against="reference"  4 x (round 5): the compiled reference's OpenBLAS sums are blocked like the GPU's tree -- measured
                           GPU-vs-reference 1.3e-13 at 512^3 / k = 3 and 5.8e-14 at 4096^2 / k = 10 (BENCH_r04), where
                           this gives 3.1e-11 / 3.6e-11: an error of 1e-10 in a fused update fails it.
    north_star's 1e-12 is what the golden-size cases (n <= 3e5) are held to; these are the size-dependent forms of the
    same bar."""
    factor = {"oracle": 32.0, "reference": 4.0}[against]
    return factor * k * float(np.sqrt(n)) * 2.220446049250313e-16  # eps = 2^-52




def _ref_solve(O, name, A, b, k, dinv, threads=1):
    """k iterations (tol = 0) by the reference's own compiled kernel: pysparse/itsolvers/src/{pcg,minres}.c unmodified
    (oracle/_ref/libref_krylov.so, refk_solve); threads > 1: row-parallel operator callbacks, same bits per row"""
    x = np.zeros(A.shape[0])
    info, it, rr, _ = O.ref_krylov(name, A, b, x, 0.0, k, ("jacobi", dinv), threads=threads)
    return x, (info, it, rr)


def gpu_parity_case(dev, O, grid, k, A=None, b=None, x_pcg=None, res_pcg=None, with_ref=True, x_ref=None, res_ref=None,
                    with_oracle=True, ref_threads=1, form="csr"):
    """`parity_check` of the bench line and tests/test_gpu_reference_sizes.py: k iterations (tol = 0) of Jacobi-PCG and
    Jacobi-MINRES on the GPU (through the host-pointer entry points, as the drop-in modules call them) against
      * the oracle's iterates on the same system (with_oracle; bound 32 k sqrt(n) eps),
      * the compiled reference where oracle/_ref was built (bound 4 k sqrt(n) eps): PCG by examples/poisson_test/pcg.c
        (libref_pcg.so) AND by the module's own pcg.c; MINRES by the module's own minres.c (libref_krylov.so, round 5).
    A / b / x_pcg / res_pcg / x_ref / res_ref: what the CPU leg already holds for this grid (else they are made here).
    with_oracle=False, ref_threads > 1: the long leg (k = 20 at 512^3) -- the sequential oracle is skipped and the compiled
    kernels get row-parallel callbacks so that it stays under a minute."""
    if A is None:
        A = O.poisson_csr(*grid)
    n = A.shape[0]
    if b is None:
        b = np.empty(n)
        A.matvec(np.ones(n), b)
    dinv = np.full(n, 1.0 / (6.0 if grid[2] else 4.0))
    x_min = res_min = None
    if with_oracle:
        if x_pcg is None:
            x_pcg = np.zeros(n)
            res_pcg = O.pcg(A, b, x_pcg, 0.0, k, dinv)
        x_min = np.zeros(n)
        res_min = O.minres(A, b, x_min, 0.0, k, dinv)
    # form "sss": the GPU side holds the operator as an sss_mat (examples/poisson_test.py: S = L.to_sss()) -- its product adds
    # a row in the csr_mat's order (sss_mat.c:45-55), so the CPU legs stay the same; its dot partials are ordered differently
    G = dev.DeviceSSS.poisson(*grid) if form == "sss" else dev.DeviceCSR.poisson(*grid)
    K = dev.DeviceJacobi(G)
    bound, bound_ref = parity_bound(n, k), parity_bound(n, k, "reference")
    out = {"grid": list(grid), "n": n, "k": k, "form": form, "bound": bound, "bound_vs_reference": bound_ref}
    ok = True
    have_refk = with_ref and O.have_ref_krylov()
    for name, solver, ref_res, ref_x in (("pcg", dev.pcg, res_pcg, x_pcg), ("minres", dev.minres, res_min, x_min)):
        xg = np.zeros(n)
        rg = solver(G, b, xg, 0.0, k, K)
        rec = {"info_iter_gpu": [rg[0], rg[1]], "ok": True}
        if with_oracle:
            rec.update({"info_iter_oracle": [ref_res[0], ref_res[1]], "relres_rel_diff": rel_diff(rg[2], ref_res[2]),
                        "x_max_rel_diff": _maxrel(xg, ref_x)})
            rec["ok"] = bool(rec["info_iter_gpu"] == rec["info_iter_oracle"] and rec["relres_rel_diff"] <= bound
                             and rec["x_max_rel_diff"] <= bound)
        if name == "pcg" and with_ref and O.have_ref() and (x_ref is not None or ref_threads == 1):
            xr, rr = x_ref, res_ref
            if xr is None:
                xr = np.zeros(n)
                rr = O.ref_pcg(A, b, xr, 0.0, k, dinv)
            v = {"kernel": "examples/poisson_test/pcg.c", "info_iter_reference": [rr[0], rr[1]],
                 "relres_rel_diff": rel_diff(rg[2], rr[2]), "x_max_rel_diff": _maxrel(xg, xr)}
            if with_oracle:
                v["oracle_vs_reference_x_max_rel_diff"] = _maxrel(ref_x, xr)
            rec["vs_reference_pcg"] = v
            rec["ok"] = bool(rec["ok"] and [rr[0], rr[1]] == rec["info_iter_gpu"] and v["x_max_rel_diff"] <= bound_ref
                             and v["relres_rel_diff"] <= bound_ref)
            del xr
        if have_refk:
            t = time.perf_counter()
            xr, rr = _ref_solve(O, name, A, b, k, dinv, ref_threads)
            v = {"kernel": "pysparse/itsolvers/src/%s.c" % name, "info_iter_reference": [rr[0], rr[1]],
                 "relres_rel_diff": rel_diff(rg[2], rr[2]), "x_max_rel_diff": _maxrel(xg, xr),
                 "callback_threads": ref_threads, "seconds": time.perf_counter() - t}
            if with_oracle:
                v["oracle_vs_reference_x_max_rel_diff"] = _maxrel(ref_x, xr)
            rec["vs_reference_module_kernel"] = v
            rec["ok"] = bool(rec["ok"] and [rr[0], rr[1]] == rec["info_iter_gpu"] and v["x_max_rel_diff"] <= bound_ref
                             and v["relres_rel_diff"] <= bound_ref)
            del xr
        ok = ok and rec["ok"]
        out[name] = rec
        del xg
    K.close()
    G.close()
    out["ok"] = ok
    return out


def _cpu_case(O, grid, spmv_reps, pcg_iters, with_ref, dev=None, threads=1):
    """oracle (C restatement of csr_mat.c:49-54 + pcg.c, gcc -O2, ONE thread) and, when it was built,
    the compiled reference PCG (oracle/_ref/libref_pcg.so = examples/poisson_test/pcg.c unmodified)"""
    t0 = time.perf_counter()
    A = O.poisson_csr(*grid)
    gen_s = time.perf_counter() - t0
    n, nnz = A.shape[0], A.nnz
    x = np.random.default_rng(0).standard_normal(n)
    y = np.empty(n)
    A.matvec(x, y)  # warm
    ts = []
    for _ in range(spmv_reps):
        t = time.perf_counter()
        A.matvec(x, y)
        ts.append(time.perf_counter() - t)
    t_spmv = float(np.median(ts))
    mt = None
    if threads > 1:
        # labelled NON-reference (optional second line): the reference's product is one thread; here the same
        # rows go to POSIX threads in contiguous ranges (oracle: orc_csr_matvec_threads), each row summed as before
        y2 = np.empty(n)
        ran = A.matvec_threads(x, y2, threads)  # warm (first touch of y2)
        tm = []
        for _ in range(max(spmv_reps, 5)):
            t = time.perf_counter()
            A.matvec_threads(x, y2, threads)
            tm.append(time.perf_counter() - t)
        mt = {"threads": ran, "spmv_ms": float(np.median(tm)) * 1e3,
              "spmv_GBps": csr_model_bytes(n, nnz) / float(np.median(tm)) / 1e9,
              "same_bits_as_one_thread": bool(np.array_equal(y, y2))}
        del y2
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    dinv = np.full(n, 1.0 / (6.0 if grid[2] else 4.0))  # jacobi(A, 1.0, 1) of the constant diagonal
    xs = np.zeros(n)
    t = time.perf_counter()
    res = O.pcg(A, b, xs, 0.0, pcg_iters, dinv)
    # iterations that really ran (a small system reaches r = 0 exactly and leaves with -2 / -5 before pcg_iters) + the
    # initial residual SpMV
    t_pcg = (time.perf_counter() - t) / (min(res[1], pcg_iters) + 1)
    out = {"n": n, "nnz": nnz, "spmv_GBps": csr_model_bytes(n, nnz) / t_spmv / 1e9, "spmv_ms": t_spmv * 1e3,
           "pcg_iters_per_s": 1.0 / t_pcg, "generate_s": gen_s,
           "sample": "median of %d SpMV; %d Jacobi-PCG iterations (tol 0)" % (spmv_reps, min(res[1], pcg_iters))}
    if mt:
        out["all_cores_non_reference"] = mt
    xr = rres = None
    if with_ref and O.have_ref():
        xr = np.zeros(n)
        t = time.perf_counter()
        rres = O.ref_pcg(A, b, xr, 0.0, pcg_iters, dinv)
        out["reference_pcg_iters_per_s"] = (min(res[1], pcg_iters) + 1) / (time.perf_counter() - t)
        # same algorithm, different BLAS-1 (OpenBLAS kernels vs the port's serial loops): the two dot products
        # of n terms differ by ~sqrt(n) eps relative, so the iterates agree to that, not to the bit
        diff = float(np.abs(xr - xs).max() / max(np.abs(xs).max(), 1e-300))
        out["reference_pcg_max_rel_diff_vs_port"] = diff
        out["reference_pcg_matches_port"] = bool(diff <= REF_PCG_TOL)
    if dev is not None and res[1] == pcg_iters + 1:
        # the same k iterations on the GPU against the iterates this leg already holds (`parity_check` of the line)
        out["gpu_parity"] = gpu_parity_case(dev, O, grid, pcg_iters, A=A, b=b, x_pcg=xs, res_pcg=res, with_ref=with_ref,
                                            x_ref=xr, res_ref=rres)
    return out


def published_table_leg(O, sizes=(100, 300, 500)):
    """The reference's ONLY published benchmark for this path -- doc/pysparse/source/itsolvers.rst:120-130 (script), :189-199
    (table): L x = 1 on the 2-D Poisson matrix, n = 100 / 300 / 500, `krylov.pcg(L.to_sss(), b, x, 1e-12, 2000)`, no
    preconditioner, seconds for assembly and solve -- reproduced end to end through the drop-in modules (ll_mat assembly on
    the host as the script does it, to_sss(), upload, the solve, x back in the caller's NumPy array), beside the reference's own
    native twin (examples/poisson_test/poisson_test.c:110-125, compiled unmodified: oracle/_ref/poisson_test) on ONE host
    core of the same box, reading the same matrix from a file as the doc's "Native C" rows do.  Pinned by
    tests/golden/ref_published_table.json (that program's iteration counts and x; oracle/make_golden.py).
    Part of the cpu_baseline leg: the compiled reference is the thing timed beside, never the thing shipped."""
    import shutil
    import tempfile
    from pysparse_amd.itsolvers import krylov
    from pysparse_amd.tools import poisson
    gold = None
    try:
        gold = {r["n"]: r for r in json.load(open(os.path.join(ROOT, "tests", "golden", "ref_published_table.json")))["rows"]}
        gx = np.load(os.path.join(ROOT, "tests", "golden", "ref_published_table.npz"))
    except (OSError, ValueError):
        gx = None
    rows, ok = [], True
    for nn in sizes:
        n = nn * nn
        t0 = time.perf_counter()
        L = poisson.poisson2d_sym_blk(nn)
        S = L.to_sss()
        t_asm = time.perf_counter() - t0
        b, x = np.ones(n), np.zeros(n)
        t0 = time.perf_counter()
        info, it, relres = krylov.pcg(S, b, x, 1e-12, 2000)
        t_first = time.perf_counter() - t0  # includes whatever the handle builds on its first solve
        x2 = np.zeros(n)
        t0 = time.perf_counter()
        info2, it2, _ = krylov.pcg(S, b, x2, 1e-12, 2000)
        t_solve = time.perf_counter() - t0
        row = {"n": nn, "rows": n, "info": info, "iter": it, "relres": relres, "gpu_assembly_s": t_asm,
               "gpu_first_solve_s": t_first, "gpu_solve_s": t_solve, "gpu_total_s": t_asm + t_first,
               "same_bits_second_solve": bool((info, it) == (info2, it2) and np.array_equal(x, x2))}
        if gold and nn in gold:
            g = gold[nn]["compiled_pcg_with_sss_product"]
            row["iter_reference_openblas"] = g["iter"]
            row["x_max_rel_diff_vs_reference"] = float(np.abs(x[::97] - gx["x_%d" % nn]).max() / np.abs(gx["x_%d" % nn]).max())
            row["ok"] = bool(info == 0 and relres <= 1e-12 and row["x_max_rel_diff_vs_reference"] <= 1e-12)
            ok = ok and row["ok"]
        if O.have_ref():
            # the unmodified native program, one core: it reads matrices/poi2d_100.mtx (name fixed in its source), converts
            # COO -> SSS and solves; wall time of the whole process = the doc's "Native C" Total
            So = O.poisson_sss(nn, nn)
            td = tempfile.mkdtemp()
            try:
                os.makedirs(os.path.join(td, "matrices"))
                with open(os.path.join(td, "matrices", "poi2d_100.mtx"), "w") as f:
                    f.write("%%%%MatrixMarket matrix coordinate real symmetric\n%d %d %d\n" % (n, n, So.nnz_lower + n))
                    # row by row: a row's lower entries, then its diagonal -- the order oracle/make_golden.py wrote the pinned
                    # files in (the program's COO -> SSS lists use 0 as "no entry": the file's FIRST entry must be a diagonal
                    # one, and a row's columns come out in reverse file order, which its own product does not mind)
                    rr = np.repeat(np.arange(n), np.diff(So.ind))
                    tot = So.nnz_lower + n
                    ii, jj, vv = np.empty(tot, dtype=np.int64), np.empty(tot, dtype=np.int64), np.empty(tot)
                    pl = np.arange(So.nnz_lower) + rr
                    pd = np.asarray(So.ind[1:], dtype=np.int64) + np.arange(n)
                    ii[pl], jj[pl], vv[pl] = rr + 1, np.asarray(So.col, dtype=np.int64) + 1, So.val
                    ii[pd], jj[pd], vv[pd] = np.arange(n) + 1, np.arange(n) + 1, So.diag
                    np.savetxt(f, np.column_stack([ii, jj, vv]), fmt="%d %d %.17g")
                t0 = time.perf_counter()
                out = subprocess.run([O.REF_BIN_PATH], cwd=td, capture_output=True, text=True,
                                     env=dict(os.environ, OPENBLAS_NUM_THREADS="1")).stdout
                wall = time.perf_counter() - t0
                row["ref_stdout_tail"] = out.strip().splitlines()[-1] if out.strip() else ""
                m = re.search(r"converged at iteration (\d+)", row["ref_stdout_tail"])
                if m:  # (a run that did not get there is no timing)
                    row["ref_total_s"] = wall
                    row["ref_iter_this_box"] = int(m.group(1))
            finally:
                shutil.rmtree(td, ignore_errors=True)
        rows.append(row)
        del L, S
    return {"what": "L x = 1, L = poisson2d_sym_blk(n).to_sss(), x0 = 0, krylov.pcg(S, b, x, 1e-12, 2000), no preconditioner: "
                    "doc/pysparse/source/itsolvers.rst:120-130,189-199; native twin examples/poisson_test/poisson_test.c",
            "rows": rows, "ok": ok if gold else None,
            "published_seconds_total": {"Python": {"100": 1.15, "300": 49.86, "500": 300.01},
                                        "Native C": {"100": 1.26, "300": 51.52, "500": 299.53},
                                        "note": "the doc's own numbers, machine unknown: context only"},
            "iteration_counts": "at tol 1e-12 the recurred residual stagnates near its floor and the count depends on the "
                                "ORDER of the BLAS-1 sums: the reference's own program gives 225 / 677 / 1132 linked with "
                                "OpenBLAS and 225 / 735 / 1297 with a sequential BLAS-1 (DESIGN.md section 7); x agrees to "
                                "1e-13 either way, which is what `ok` checks"}


def _usable_cores():
    """threads the "all host cores" line may use: the affinity mask, cut to the cgroup's CPU quota where one is set (a GPU
    box shows all 256 hardware threads to a job that owns 16 of them) and to 64 (one thread's ranges stay >= 2 MB at C2)"""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    quota = None
    try:  # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(per)
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        cores = min(cores, max(1, int(quota + 0.5)))
    return max(1, min(cores, 64))


def cpu_baseline(budget_s=75.0, c2_grid=(4096, 4096, 0), c3_grid=(512, 512, 512), c3_small=(256, 256, 256), dev=None,
                 published=False):
    """`cpu_baseline` (kind "port": the oracle's SpMV -- the reference's csr_mat.c needs the Python-2
    C API and cannot be compiled) and `cpu_baseline_reference_pcg` (kind "reference": the reference's
    own pcg.c, compiled unmodified, driven by the oracle's CSR matvec callback).  Sizes: C2 (4096^2)
    always; C3 (512^3, 14 GB of matrix) when host memory and the time budget allow, else 256^3."""
    from oracle import oracle as O
    t0 = time.time()
    c1 = _cpu_case(O, (100, 100, 0), 200, 2000, True)  # configs[0]: the reference's own CPU-runnable case
    cores = _usable_cores()
    c2 = _cpu_case(O, c2_grid, 10, 10, True, dev, cores)
    # C3 costs ~11x C2's generation + ~10x its per-pass time
    predicted = 11.2 * c2["generate_s"] + 10.5 * (5 * c2["spmv_ms"] * 1e-3 + 2 * 5 / c2["pcg_iters_per_s"])
    big = _mem_available_gb() > 48 and predicted < budget_s
    grid3 = c3_grid if big else c3_small
    c3 = _cpu_case(O, grid3, 3 if big else 5, 3 if big else 10, True, dev, cores)
    model, nproc = _cpu_model(), os.cpu_count()
    head = c3 if big else c2
    base = {
        "value": head["spmv_GBps"], "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": "CSR SpMV, 7-pt Poisson %d^3 (n=%d, nnz=%d): %s; C2 and C3 below"
                  % (grid3[0], c3["n"], c3["nnz"], c3["sample"]) if big else
                  "CSR SpMV, 5-pt Poisson %d^2 (n=%d, nnz=%d): %s; %d^3 skipped (host memory / time budget), "
                  "%d^3 below" % (c2_grid[0], c2["n"], c2["nnz"], c2["sample"], c3_grid[0], c3_small[0]),
        "pcg_iters_per_s": head["pcg_iters_per_s"], "host_cpu": model, "host_nproc": nproc,
        "C1_poisson2d_100": c1,
        "C2_poisson2d_%d" % c2_grid[0]: c2, ("C3_poisson3d_%d" % c3_grid[0] if big else "poisson3d_%d" % c3_small[0]): c3,
        "seconds": time.time() - t0,
    }
    ref = None
    if "reference_pcg_iters_per_s" in head:
        ref = {"value": head["reference_pcg_iters_per_s"], "unit": "PCG iterations/s", "cores": 1,
               "kind": "reference",
               "sample": "examples/poisson_test/pcg.c compiled unmodified (oracle/_ref/libref_pcg.so, BLAS-1 from "
                         "OpenBLAS with 1 thread), Jacobi-PCG on %s, %s" % (
                             "7-pt Poisson %d^3" % c3_grid[0] if big else "5-pt Poisson %d^2" % c2_grid[0], head["sample"]),
               "C2_poisson2d_%d" % c2_grid[0]: c2.get("reference_pcg_iters_per_s"),
               ("C3_poisson3d_%d" % c3_grid[0] if big else "poisson3d_%d" % c3_small[0]): c3.get("reference_pcg_iters_per_s"),
               "iterates_match_port": bool(c2.get("reference_pcg_matches_port")
                                           and c3.get("reference_pcg_matches_port")),
               "iterates_tolerance": REF_PCG_TOL,
               "iterates_max_rel_diff": max(c2.get("reference_pcg_max_rel_diff_vs_port", 0.0),
                                            c3.get("reference_pcg_max_rel_diff_vs_port", 0.0))}
    parity = None
    if dev is not None:
        cases = {name: c.pop("gpu_parity") for name, c in (("C2_poisson2d_%d" % c2_grid[0], c2),
                                                           (("C3_poisson3d_%d" % c3_grid[0]) if big else
                                                            "poisson3d_%d" % c3_small[0], c3)) if "gpu_parity" in c}
        parity = {"what": "k iterations (tol = 0) of Jacobi-PCG and Jacobi-MINRES on the GPU against the oracle's iterates "
                          "of the same system (b = A*ones, x0 = 0), and against the reference's own compiled kernels "
                          "(examples/poisson_test/pcg.c, pysparse/itsolvers/src/pcg.c and minres.c) where oracle/_ref "
                          "exists: equal (info, iter); relres and max-norm of x within `bound`",
                  "bound": "GPU vs oracle: 32 k sqrt(n) eps; GPU vs the compiled reference (pcg.c, minres.c): 4 k sqrt(n) eps; "
                           "eps = 2^-52 (bench.parity_bound; DESIGN.md section 7)",
                  "cases": cases, "ok": bool(cases) and all(c["ok"] for c in cases.values())}
    if published and dev is not None:
        try:
            base["published_table"] = published_table_leg(O)
        except Exception as e:  # noqa: BLE001 - a reported extra, never fatal for the bench line
            base["published_table"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    return base, ref, parity






# ------------------------------------------------------------------------------------ launcher



























def mtx_leg(spec, steps=50, minres_iters=200):
    """BASELINE.json configs[4] (an unstructured symmetric matrix as sss_mat + MINRES; the reference's flow is
    examples/demo_pcg.py:47-98 over a MatrixMarket file through ll_mat.c:3390-3456) for a matrix the user supplies:
    spec = a .mtx path, or `standin:fem32` / `standin:fem512` / `standin:logspaced` (pysparse_amd/tools/standins.py; the
    SuiteSparse file cannot be fetched here).  Reports: ingest (parse + sort + split into sss arrays) and upload times,
    the time to the first product (tables, renumbering), the kernel chosen, SpMV time priced in SSS-model bytes
    (12 nnz_lower + 28 n + 4) and in CSR-model bytes of the full matrix, Jacobi-MINRES microseconds per
    iteration, and parity against the oracle: the product bit for bit, MINRES info / iterations / iterate."""
    from pysparse_amd import _capi, device as dev
    from oracle import oracle as O
    L, check = _capi.lib(), _capi.check

    def sync():
        check(L.psp_synchronize())
    t0 = time.perf_counter()
    if spec.startswith("standin:"):
        from pysparse_amd.tools import standins
        kind = spec.split(":", 1)[1]
        if kind == "logspaced":
            n, ind, col, val, diag = standins.logspaced_sss_arrays(923136)
        else:
            n, ind, col, val, diag = standins.fem_sss_arrays(68, 68, 67, int(kind[3:] or 32))
        source = "seeded stand-in %s (pysparse_amd/tools/standins.py), NOT the SuiteSparse file" % kind
    else:
        from pysparse_amd.tools import mtx
        with open(spec, "r") as f:
            banner = f.readline().lower().split()
        if len(banner) < 5 or banner[4] != "symmetric":
            # to_sss() of a general matrix silently drops its upper triangle (ll_mat.c:1654-1708): numbers for half a matrix
            raise SystemExit("--mtx expects a SYMMETRIC coordinate file (configs[4] is an sss_mat); banner: " + " ".join(banner))
        n, ind, col, val, diag = mtx.sss_arrays_from_mtx(spec)
        source = "MatrixMarket file " + os.path.basename(spec)
    ingest_s = time.perf_counter() - t0
    nnz_lower = int(val.shape[0])
    rng = np.random.default_rng(7)
    xh = rng.standard_normal(n)
    b = np.zeros(n)
    b[0] = 1.0
    b += 1e-3 * rng.standard_normal(n)
    # a process pays for its first large host-to-device copy (~160 ms) and for the first launch from each translation unit
    # (~15 ms each) whatever the matrix: out of the way before anything is timed, on a 3 000-row stand-in of the same kind
    from pysparse_amd.tools import standins as _st
    wn, wi, wc, wv, wd = _st.fem_sss_arrays(10, 10, 10, 64)
    Sw = dev.DeviceSSS.from_arrays(wn, wi, wc, wv, wd)
    Sw.prepare(1 << 30)
    Kw = dev.DeviceJacobi(Sw)
    dev.minres(Sw, np.ones(wn), np.zeros(wn), 1e-10, 50, Kw)
    Kw.close()
    Sw.close()
    dev.DeviceBuffer.from_host(np.ones(1 << 22)).free()
    sync()
    # ---- time to solution, as a script that solves ONCE sees it (examples/demo_pcg.py:47-98): arrays up, Jacobi, converged
    # MINRES with x back in the caller's array.  The handle decides by its cost rule what to build (psp_csr.hip
    # pick_scattered: an irregular numbering multiplies with csr_spmv_w5 and gets its renumbered copy after 2048 products)
    t_all = time.perf_counter()
    t0 = time.perf_counter()
    S = dev.DeviceSSS.from_arrays(n, ind, col, val, diag)
    sync()
    upload_s = time.perf_counter() - t0
    K = dev.DeviceJacobi(S)
    xg, xo = np.zeros(n), np.zeros(n)
    t0 = time.perf_counter()
    got = dev.minres(S, b, xg, 1e-10, 500, K)
    first_solve_s = time.perf_counter() - t0
    end_to_end_s = time.perf_counter() - t_all
    kern0, kinfo0 = S.kernel_info()
    So = O.SSS(n, val, diag, col, ind)
    ref = O.minres(So, b, xo, 1e-10, 500, O.jacobi_dinv(diag))
    x_diff = _maxrel(xg, xo)
    t0 = time.perf_counter()
    dev.minres(S, b, np.zeros(n), 1e-10, 500, K)
    second_solve_s = time.perf_counter() - t0

    def us_per_iteration():
        times = {}
        for k in (20, 20 + minres_iters):  # the difference cancels the transfers of b and x and the set-up product
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                dev.minres(S, b, np.zeros(n), 0.0, k, K)
                best = min(best, time.perf_counter() - t0)
            times[k] = best
        return (times[20 + minres_iters] - times[20]) / minres_iters * 1e6
    xb, yb = dev.DeviceBuffer.from_host(xh), dev.DeviceBuffer(n)
    ev = Events(L, check, steps + 1)
    cold = {"kernel": kern0, "kernel_info": kinfo0}
    timed_launches(lambda: S.matvec_dev(xb.ptr, yb.ptr), sync, ev, 5)
    cold["spmv_ms"], _ = timed_launches(lambda: S.matvec_dev(xb.ptr, yb.ptr), sync, ev, steps)
    cold["minres_us_per_iteration"] = us_per_iteration()
    cold["products_counted"] = S.setup_info()["products_counted"]
    # ---- steady state: the caller announces many products (psp_sss_prepare), the handle builds what pays for itself
    S.prepare(1 << 30)
    t0 = time.perf_counter()
    S.matvec_dev(xb.ptr, yb.ptr)  # builds the renumbered copy and its tables where the numbering is irregular
    sync()
    first_product_s = time.perf_counter() - t0
    kern, kinfo = S.kernel_info()
    setup = S.setup_info()
    timed_launches(lambda: S.matvec_dev(xb.ptr, yb.ptr), sync, ev, 5)
    avg, med = timed_launches(lambda: S.matvec_dev(xb.ptr, yb.ptr), sync, ev, steps)
    sss_bytes = 12 * nnz_lower + 28 * n + 4
    csr_bytes = csr_model_bytes(n, 2 * nnz_lower + n)
    # parity: the product against the oracle's sss_matvec loop (sss_mat.c:40-56), bit for bit
    yo = np.empty(n)
    So.matvec(xh, yo)
    yg = yb.download()
    spmv_bits = bool(np.array_equal(yg, yo))
    xg2 = np.zeros(n)
    got2 = dev.minres(S, b, xg2, 1e-10, 500, K)  # in the copy's numbering now
    x_diff2 = _maxrel(xg2, xo)
    us_iter = us_per_iteration()
    out = {"source": source, "n": n, "nnz_lower": nnz_lower, "nnz_full": 2 * nnz_lower + n,
           "ingest_s": ingest_s, "upload_s": upload_s, "first_solve_s": first_solve_s, "end_to_end_s": end_to_end_s,
           "second_solve_s": second_solve_s, "first_product_s": first_product_s,
           "time_to_solution": {"what": "arrays up + Jacobi + converged MINRES (tol 1e-10) + x on the host, warm process, the "
                                        "handle's own cost rule", "end_to_end_ms": end_to_end_s * 1e3,
                                "upload_ms": upload_s * 1e3, "first_solve_ms": first_solve_s * 1e3,
                                "second_solve_ms": second_solve_s * 1e3, "kernel": kern0,
                                "round5_same_flow_ms": "upload 116-120 + first product 47-57 + solve 5-6 "
                                                       "(profiles/r5_mtx_leg_standins.jsonl; first large copy of the process "
                                                       "included there)"},
           "cold": cold, "setup_ms": setup["reorder_ms"], "setup": setup,
           "kernel": kern, "kernel_info": dict(kinfo, setup_ms=setup["reorder_ms"]), "spmv_ms": avg, "spmv_median_ms": med,
           "sss_model_bytes": sss_bytes, "sss_model_GBps": sss_bytes / (avg * 1e-3) / 1e9,
           "sss_model_frac_of_peak": sss_bytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "csr_model_bytes": csr_bytes, "csr_model_GBps": csr_bytes / (avg * 1e-3) / 1e9,
           "csr_model_frac_of_peak": csr_bytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "minres": {"info": got[0], "iter": got[1], "relres": got[2], "solve_s_with_transfers": first_solve_s,
                      "us_per_iteration": us_iter, "iters_timed": minres_iters},
           "parity": {"spmv_bit_exact_vs_oracle": spmv_bits, "minres_info_iter_oracle": [ref[0], ref[1]],
                      "minres_info_iter_gpu": [got[0], got[1]], "x_max_rel_diff": x_diff,
                      "minres_info_iter_gpu_renumbered": [got2[0], got2[1]], "x_max_rel_diff_renumbered": x_diff2,
                      "x_tol": 1e-12,
                      "ok": bool(spmv_bits and (got[0], got[1]) == (ref[0], ref[1]) == (got2[0], got2[1])
                                 and x_diff <= 1e-12 and x_diff2 <= 1e-12)}}
    K.close()
    S.close()
    xb.free()
    yb.free()
    return out


def mtx_main(a):
    """`python bench.py --mtx PATH|standin:NAME`: the configs[4] leg on its own, one JSON line"""
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    leg = mtx_leg(a.mtx, steps=a.steps)
    from pysparse_amd import _capi
    out = {"metric": "sss_mat SpMV GB/s in SSS-model bytes (% of 8 TB/s HBM peak) + Jacobi-MINRES us/iteration",
           "value": leg["sss_model_GBps"], "unit": "GB/s", "n_gpus": 1, "steps": a.steps, "warmup": 5,
           "ms_per_step": leg["spmv_ms"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "user file" if not a.mtx.startswith("standin:") else "synthetic stand-in",
           "config": {"workload": "BASELINE.json configs[4]: unstructured symmetric matrix as sss_mat, y = S x and "
                                  "Jacobi-MINRES on 1 GPU -- " + leg["source"], "n": leg["n"], "nnz": leg["nnz_full"]},
           "roofline": {"bound": "hbm", "kernel": leg["kernel"], "achieved": leg["sss_model_GBps"], "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": leg["sss_model_frac_of_peak"], "traffic": None,
                        "algorithmic_bytes_per_launch": leg["sss_model_bytes"], "avg_launch_ms": leg["spmv_ms"],
                        "note": "SSS model of SURVEY 8d (12 nnz_lower + 28 n + 4); an irregular sss_mat multiplies with its "
                                "expanded mirror, which moves about twice that (DESIGN.md section 2): csr_model_* prices "
                                "the same time in the full matrix's CSR bytes"},
           "config5": leg, "provenance": provenance(_capi.lib())}
    if not leg["parity"]["ok"]:
        out["error"] = "parity against the oracle failed"
    out["stage"] = "mtx"
    emit(out, real_stdout, a.side_file or None)
    return 1 if "error" in out else 0






def main():
    import signal
    signal.pthread_sigmask(signal.SIG_UNBLOCK, {signal.SIGTERM})  # (a fall-back stage started by a guarded rank inherits its mask)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pcg-iters", type=int, default=100)
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="auto: N = 1 -> the 512^3 single-GPU workload, N > 1 -> strong (fixed 1024^3)")
    ap.add_argument("--grid", default="", help="override the grid, e.g. 256,256,256 (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sss", action="store_true", help="skip the sss_mat leg (N = 1)")
    ap.add_argument("--no-solvers", action="store_true", help="skip the cgs / bicgstab / qmrs / gmres(20) leg (N = 1)")
    ap.add_argument("--no-kernels", action="store_true", help="skip the w3 / w2 legs on the same operator (N = 1)")
    ap.add_argument("--no-strong-n1", action="store_true", help="skip the one-GPU 1024^3 leg")
    ap.add_argument("--no-clocks", action="store_true")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child runs (N = 1); the committed "
                         "passes under profiles/ are quoted instead")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (with --backend gloo, which moves device "
                         "tensors on this image; RCCL refuses two ranks on one GPU).  Not a measurement.")
    ap.add_argument("--test-backend", default="",
                    help="module:factory returning (backend, make_local) -- CPU dry run of the launcher and the "
                         "row-range driver over gloo (tests/); the line it prints is marked dry_run, not a measurement")
    ap.add_argument("--force-dist", action="store_true",
                    help="use the torch.distributed driver even at world size 1 (plumbing check)")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs from ONE process through the C ABI's device-list variant (psp_csr_poisson_multi) "
                         "instead of one torch.distributed rank per GPU")
    ap.add_argument("--ladder", default="",
                    help="N > 1 as a plain script: comma-separated stages to try in order (default: %s)" % ",".join(LADDER))
    ap.add_argument("--deadline", type=float, default=540.0,
                    help="N > 1 as a plain script: seconds the whole ladder may take; a line (result or error) is "
                         "printed before it passes")
    ap.add_argument("--stage-timeout", type=float, default=0.0,
                    help="N > 1 as a plain script: time-out of every stage in seconds (default: per-stage caps inside "
                         "--deadline)")
    ap.add_argument("--stage", default="", help="set by the ladder: which stage this process is")
    ap.add_argument("--rank-deadline", type=float, default=270.0,
                    help="one rank of N started by an external torch.distributed.run: seconds after which the rank's "
                         "watchdog gives up on the torch / RCCL path (rank 0 then runs the single-process stages)")
    ap.add_argument("--inject", default="",
                    help="failure injection for the launcher tests: exit:RANK (that rank leaves with code 3 after the "
                         "process group formed) or hang:RANK (that rank sleeps instead of taking part)")
    ap.add_argument("--no-phases", action="store_true", help="skip the per-phase timing of an N > 1 iteration")
    ap.add_argument("--side-file", default="",
                    help="where the full record goes (default gpurun_out/bench_side_n<N>.json under the repo); the "
                         "printed line stays <= 6 KB and names it as `side_file`")
    ap.add_argument("--mtx", default="",
                    help="configs[4] on its own: a symmetric MatrixMarket file (e.g. SuiteSparse Emilia_923.mtx) or "
                         "standin:fem32 | standin:fem512 | standin:logspaced -- sss_mat product + Jacobi-MINRES on one "
                         "GPU with parity against the oracle.  The default run adds the same object as `config5` when "
                         "the environment variable EMILIA_MTX names a file")
    a = ap.parse_args()

    if a.mtx:
        raise SystemExit(mtx_main(a))

    if a.single_process:
        raise SystemExit(single_process_main(a))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        argv = [t for t in sys.argv[1:]]
        raise SystemExit(orchestrate(a, argv))

    # ONE JSON line on stdout: RCCL / HIP print banners to file descriptor 1, so everything this process
    # (and the libraries it loads) writes to stdout goes to stderr, and rank 0's line to the real stdout
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not a.stage:
        # started as one rank of N by somebody else's `torch.distributed.run` (the driver's scaling run), not by the
        # ladder above: this rank guards itself
        return guarded_rank(a, real_stdout, run_body)
    return run_body(a, real_stdout)






def timed_region(step, sync, ev, warmup, steps):
    """THE timed region of the contract.  `warmup` untimed steps, then exactly `steps` steps bracketed by sync() on both
    sides (N = 1: psp_synchronize; N > 1: device synchronise + barrier + device synchronise).  Returns (wall seconds of the
    K steps, milliseconds between two HIP events recorded around the same K launches on the stream the library launches
    on -- the wall time in ms when there is no device, i.e. in the gloo dry runs).  Nothing else happens in here: no
    allocation, no side leg, no oracle (tests/test_bench_region.py)."""
    for _ in range(warmup):
        step()
    sync()
    t0 = time.perf_counter()
    if ev:
        ev.record(0)
    for _ in range(steps):
        step()
    if ev:
        ev.record(1)
    sync()
    wall = time.perf_counter() - t0
    return wall, (ev.ms(0, 1) if ev else wall * 1e3)


def headline(kbytes_job, kbytes_launch, wall, ev_ms, steps):
    """`value` / `ms_per_step` / `roofline.achieved` from what timed_region measured: value = bytes the kernels of the
    WHOLE JOB have to move per step / wall time per step (max over ranks); achieved = one GPU's bytes per launch / that
    kernel's average launch over the same K launches (HIP events)"""
    kern_ms = ev_ms / steps
    achieved = kbytes_launch / (kern_ms * 1e-3) / 1e9
    return {"value": kbytes_job / (wall / steps) / 1e9, "ms_per_step": wall * 1e3 / steps, "avg_launch_ms": kern_ms,
            "achieved": achieved, "frac": achieved / HBM_PEAK_GBPS}


def run_body(a, real_stdout):

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world

    scaling = a.scaling
    if scaling == "auto":
        scaling = "strong" if world > 1 else "single"
    use_dist = world > 1 or a.force_dist or scaling != "single"
    if a.test_backend:
        use_dist = True
    if a.grid:
        nx, ny, nz = (int(t) for t in a.grid.split(","))
    elif scaling == "single":
        nx = ny = nz = 512
    elif scaling == "strong":
        nx = ny = nz = 1024
    else:
        nx = ny = 1024
        nz = 128 * world

    # roofline.traffic from the hardware counters of this job's own runs -- child processes, and before this
    # process creates its GPU context (see live_traffic)
    pmc_live, pmc_reason = None, None
    if world == 1 and not use_dist and not a.no_pmc:
        pmc_live, pmc_reason = live_traffic((nx, ny, nz), a.variant)

    if use_dist or os.environ.get("PSP_IMPORT_TORCH"):
        # torch first: its bundled HIP runtime must be the one libpysparse_hip.so binds to --
        # two HIP runtimes in one process do not both see the GPU
        import torch
        import torch.distributed as dist
    from pysparse_amd import _capi, device as dev
    L, check = _capi.lib(), _capi.check

    strong_n1 = None
    exit_code = 0
    dry = bool(a.test_backend)
    if use_dist:
        from pysparse_amd import distributed as D
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if dry:
            import importlib
            mod, fn = a.test_backend.split(":")
            dist.init_process_group("gloo")
            be, make_local = getattr(importlib.import_module(mod), fn)()
            dev_sync = lambda: None  # noqa: E731
        else:
            if a.share_gpu:
                local_rank = 0
            torch.cuda.set_device(local_rank)
            if a.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(a.backend)
            be = D.HipBackend(local_rank)
            make_local = dev.DeviceCSR.poisson_big_slab if scaling == "strong" else dev.DeviceCSR.poisson_slab
            dev_sync = torch.cuda.synchronize
        comm = D.Comm()
        if a.inject:  # launcher tests: a rank that dies / a rank that never arrives, after the group has formed
            kind, _, who = a.inject.partition(":")
            if int(who or -1) == rank:
                if kind == "exit":
                    print("[bench] injected failure: rank %d exits" % rank, file=sys.stderr, flush=True)
                    os._exit(3)
                if kind == "hang":
                    print("[bench] injected failure: rank %d sleeps" % rank, file=sys.stderr, flush=True)
                    time.sleep(1e6)
        preflight = None
        if rank == 0:
            preflight = {"world": world, "backend": dist.get_backend()}
            if not dry:
                ndev = L.psp_device_count()
                preflight["device_count"] = ndev
                preflight["peer_access"] = peer_matrix(L, 1 if a.share_gpu else min(ndev, world))
                if world > 1 and not a.share_gpu:
                    preflight["link_topology"] = link_topology()
        if scaling == "strong" and world > 1 and rank == 0 and not a.no_strong_n1:
            # the 1-GPU end of the strong-scaling ratio and of the parity check, measured in this job on rank 0's GPU
            if dry:
                strong_n1 = dry_strong_n1(a.test_backend, (nx, ny, nz))
            else:
                strong_n1 = strong_n1_leg(L, check, dev, (nx, ny, nz), min(a.pcg_iters, 24))
        comm.barrier()
        A = D.DistCSR.poisson(nx, ny, nz, comm, be, make_local)
        Aloc = A.A
        if not dry:
            Aloc.set_variant(a.variant)
        n_loc, nnz_loc = Aloc.shape[0], Aloc.nnz
        x_ext = A.new_ext()
        A.owned(x_ext).copy_(be.from_numpy(np.random.default_rng(rank).standard_normal(n_loc))
                             if n_loc <= (1 << 24) else
                             torch.randn(n_loc, dtype=torch.float64, device=x_ext.device,
                                         generator=torch.Generator(device=x_ext.device).manual_seed(rank)))
        y = be.zeros(n_loc)

        def step():
            A.matvec(x_ext, y)

        def sync():
            dev_sync()
            dist.barrier()
            dev_sync()
    else:
        check(L.psp_set_device(local_rank))
        A = dev.DeviceCSR.poisson(nx, ny, nz)
        A.set_variant(a.variant)
        Aloc = A
        n_loc, nnz_loc = A.shape[0], A.nnz
        xb = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n_loc))
        yb = dev.DeviceBuffer(n_loc)

        def step():
            A.matvec_dev(xb.ptr, yb.ptr)

        def sync():
            check(L.psp_synchronize())

    # ---- the timed region (timed_region above): W warm-up steps, then exactly K steps between barrier + synchronize
    ev = None if dry else Events(L, check, a.steps + 1)
    wall, ev_ms = timed_region(step, sync, ev, a.warmup, a.steps)
    if dry:
        med_ms = ev_ms
        kernel, kinfo = "test-backend", {}
    else:
        # per-launch median of the same K launches (SURVEY 8d protocol), outside the timed region
        _, med_ms = timed_launches(step, sync, ev, a.steps)
        kernel, kinfo = Aloc.kernel_info()
    kbytes_loc = kernel_bytes(kernel, kinfo, n_loc, nnz_loc)

    # ---- beside it (N = 1; bench_legs.py): the other SpMV kernels on the SAME operator, the placement sweep, the
    # streaming ceilings of this job.  Reported only -- none of them feeds `value` / `roofline.achieved`.
    kernels = placement = ceiling = None
    if not use_dist and not a.no_kernels:
        ctx = {"L": L, "check": check, "dev": dev, "A": A, "xb": xb, "yb": yb, "n": n_loc, "nnz": nnz_loc, "step": step,
               "sync": sync, "ev": ev, "steps": a.steps, "kbytes": kbytes_loc}
        if a.variant < 0:
            kernels = same_operator_kernels_leg(ctx)
        if not dry:
            placement = placement_sweep_leg(ctx)
            ceiling = stream_ceiling_leg(ctx)

    clocks = None
    if not a.no_clocks and not dry:
        # every rank keeps its GPU busy for ~1 s (N > 1: the steps exchange halos, so all ranks take part);
        # rank 0 reads the clocks meanwhile
        for _ in range(600):
            step()
        if rank == 0:
            clocks = gpu_clocks()
        sync()

    # ---- Jacobi-PCG iterations/s on the same operator (b = A*ones, x0 = 0, tol = 0 so that exactly k
    # iterations run; the setup -- ||b||, r = b - A x0 -- is inside the timed region)
    k = a.pcg_iters
    if use_dist:
        ones = A.new_ext()
        ones.fill_(1.0)
        b = be.zeros(n_loc)
        A.matvec(ones, b)
        del ones
        dinv = be.zeros(n_loc)
        dinv.fill_(1.0 / (6.0 if nz > 0 else 4.0))  # constant diagonal of the Poisson operator
        parity_mine = None
        for kk in (PARITY_ITERS, k):  # first call = warm-up and the in-job parity solve
            xs = be.zeros(n_loc)
            sync()
            t = time.perf_counter()
            res = D.dist_pcg(A, b, xs, 0.0, kk, dinv)
            sync()
            pcg_t = time.perf_counter() - t
            if parity_mine is None:
                # checksums of x after PARITY_ITERS iterations: x.b and x.x, summed over the ranks
                cs = torch.stack([be.dot(xs, b).reshape(()).clone(), be.dot(xs, xs).reshape(()).clone()])
                comm.allreduce_sum(cs)
                cs = cs.tolist()
                parity_mine = {"relres": res[2], "x_dot_b": cs[0], "x_dot_x": cs[1], "info_iter": [res[0], res[1]]}
        pcg_s_per_iter = pcg_t / k
        pcg_path = "pysparse_amd.distributed.dist_pcg (row-range driver, %s)" % D.dist_pcg_mode()
        # the row-range driver's loop is the lazy one cut at its two reductions (distributed.py): 64 n beside the product
        pcg_loop, pcg_loop_info = "dist_pcg_lazy", {"launches": 7, "vec_bytes_per_row": 64, "dinv_streamed": False,
                                                    "single_kernel_fallbacks": 0}
        # where the time of an iteration goes: marks between the phases of the same loop (events on the stream the
        # kernels and the collectives are ordered on), + the ghost exchange on its own
        phases = None
        if not a.no_phases and world > 1:
            timer = D.PhaseTimer(cuda=not dry)
            xs = be.zeros(n_loc)
            D.dist_pcg(A, b, xs, 0.0, 12, dinv, timer=timer)
            phases = timer.summary(skip=2)
            pe = A.new_ext()
            phases["halo_ms"] = A.halo_time(pe, reps=10)
            del pe
            tp = torch.tensor([phases.get(kx, 0.0) for kx in D.PhaseTimer.PHASES] + [phases["halo_ms"], phases["iteration"]],
                              dtype=torch.float64, device="cpu" if dry else "cuda")
            dist.all_reduce(tp, op=dist.ReduceOp.MAX)  # the slowest rank per phase
            tp = tp.tolist()
            phases = {kx + "_ms": tp[i] for i, kx in enumerate(D.PhaseTimer.PHASES)}
            phases["halo_ms"], phases["iteration_ms"] = tp[-2], tp[-1]
            phases["allreduce_us"] = [phases["allreduce_1_ms"] * 1e3, phases["allreduce_2_ms"] * 1e3]
            if phases["halo_ms"] > 0:
                phases["overlap_frac"] = max(0.0, min(1.0, 1.0 - phases["halo_exposed_ms"] / phases["halo_ms"]))
            phases["note"] = ("max over ranks of the mean per phase over 10 iterations; halo_exposed = what the boundary "
                              "rows still wait for after the interior rows; halo_ms = one ghost exchange on its own")
    else:
        pcg_s_per_iter, res = pcg_single(L, check, dev, A, n_loc, k, sync)
        pcg_loop, pcg_loop_info = dev.last_solve_info()  # which of the library's loops ran: the byte model follows it
        pcg_path = "psp_pcg_dev (single-GPU device-resident loop: %s)" % pcg_loop
        parity_mine, phases, preflight = None, None, None

    # ---- beside it (N = 1; bench_legs.py): the same operator as an sss_mat (examples/poisson_test.py solves with
    # S = L.to_sss()): y = S x from the strict lower triangle only, and Jacobi-PCG on it
    sss = None
    if not use_dist and not a.no_sss:
        sss = sss_leg({"L": L, "check": check, "dev": dev, "xb": xb, "yb": yb, "n": n_loc, "sync": sync, "ev": ev,
                       "steps": a.steps, "grid": (nx, ny, nz), "pcg_iters": k})

    # ---- beside it (N = 1; bench_legs.py): the reference's other Krylov solvers on the same operator
    solvers = None
    if not use_dist and not dry and not a.no_solvers:
        try:
            solvers = solvers_leg({"dev": dev, "A": A, "n": n_loc, "kbytes": kbytes_loc})
        except Exception as e:  # noqa: BLE001 - a reported extra, never fatal
            solvers = {"error": str(e)[:200]}

    # ---- beside it (N = 1 default run; bench_legs.py): the single-kernel loops at configs[0]'s size and at 10^6 points
    single_kernel = None
    if not use_dist and not dry and not a.grid and not a.no_solvers:
        try:
            single_kernel = single_kernel_leg(L, check, dev)
        except Exception as e:  # noqa: BLE001 - a reported extra, never fatal
            single_kernel = {"error": str(e)[:200]}

    # ---- N = 1 default run: the 1-GPU end of the strong-scaling target (1024^3 on this GPU)
    if not use_dist and not a.grid and not a.no_strong_n1:
        A.close()
        xb.free()
        yb.free()
        check(L.psp_trim())
        strong_n1 = strong_n1_leg(L, check, dev, (1024, 1024, 1024), min(k, 24))

    # ---- MAX over ranks
    if use_dist:
        tdev = "cpu" if dry else "cuda"
        t = torch.tensor([wall, ev_ms, pcg_s_per_iter, med_ms], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms, pcg_s_per_iter, med_ms = t.tolist()
        tot = torch.tensor([float(n_loc), float(nnz_loc), float(kbytes_loc)], dtype=torch.float64, device=tdev)
        dist.all_reduce(tot)
        n_tot, nnz_tot, kbytes_tot = (int(v) for v in tot.tolist())
    else:
        n_tot, nnz_tot, kbytes_tot = n_loc, nnz_loc, kbytes_loc

    if rank == 0:
        hl = headline(kbytes_tot, kbytes_loc, wall, ev_ms, a.steps)
        ms_step, value, kern_ms, achieved = hl["ms_per_step"], hl["value"], hl["avg_launch_ms"], hl["achieved"]
        # bytes one PCG iteration has to move: the product in the format of the kernel that ran + what the loop THAT RAN
        # (psp_last_solve_info) streams per row beside it -- 56 n for the folded lazy loop, 64 n lazy, 72 n eager (round-5
        # advisor finding: the line used to price the folded loop at 64 n)
        vb = pcg_loop_info["vec_bytes_per_row"]
        pcg_moved = kbytes_tot + (vb * n_tot if vb >= 0 else pcg_vector_bytes(n_tot, True))
        traffic, traffic_source, traffic_detail = None, None, None
        if pmc_live is not None:
            hits = [v for name, v in pmc_live.items() if kernel in name]
            if hits:
                traffic = hits[0]["bytes"]
                traffic_detail = {k: v for k, v in hits[0].items() if k != "bytes"}
                traffic_source = ("measured in this job: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two child runs of "
                                  "tools/prof_spmv.py on the same operator (L2<->fabric bytes, Infinity-Cache hits "
                                  "included; FETCH_SIZE doubled per MI355X_MICROARCH.md)")
            else:
                traffic_detail = "no counter samples of " + kernel
        elif pmc_reason:
            traffic_detail = pmc_reason
        pmc = os.path.join(ROOT, "profiles", PMC_FILES.get(kernel, ""))
        if traffic is None and os.path.isfile(pmc) and world == 1 and scaling == "single" and not a.grid:
            # HBM-side bytes of one launch from the committed rocprofv3 --pmc passes of this kernel on this
            # workload (tools/make_profiles.sh): counters cannot be read in-process, so this is NOT measured
            # in this run -- `traffic_source` says where it comes from
            try:
                rec = json.load(open(pmc))
                if rec.get("kernel") == kernel:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/" + PMC_FILES[kernel] + " (rocprofv3 --pmc, separate run%s)" % (
                        "; live measurement not available: %s" % traffic_detail if traffic_detail else "")
                    traffic_detail = None
            except (OSError, ValueError):
                traffic = None
        out = {
            "metric": METRIC,
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True,
            "scaling": "weak" if scaling == "weak" else "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "3D Poisson 7-pt %dx%dx%d fp64 csr_mat (int32 indices), y = A x%s" % (
                    nx, ny, nz, "" if world == 1 else "; z-slab row partition, ghost exchange over %s" % (
                        "RCCL" if dist.get_backend() == "nccl" else dist.get_backend())),
                "n": n_tot, "nnz": nnz_tot, "rows_per_gpu": n_loc,
                "parallelism": "1 GPU" if world == 1 else "row-range x%d" % world,
                "scaling_mode": scaling,
            },
            "value_basis": "bytes the kernel that ran has to move (its own matrix format + x + y once); "
                           "effective_csr_model_GBps = the same time priced in CSR-model bytes 12 nnz + 20 n + 4",
            "pct_hbm_peak": 100.0 * value / (HBM_PEAK_GBPS * world),
            "effective_csr_model_GBps": csr_model_bytes(n_tot, nnz_tot) / (wall / a.steps) / 1e9,
            "pcg_iters_per_s": 1.0 / pcg_s_per_iter,
            "pcg_effective_GBps": pcg_moved / pcg_s_per_iter / 1e9,
            "pcg_pct_hbm_peak": 100.0 * pcg_moved / pcg_s_per_iter / 1e9 / (HBM_PEAK_GBPS * world),
            "pcg_csr_model_equiv_GBps": (12 * nnz_tot + 108 * n_tot) / pcg_s_per_iter / 1e9,
            "pcg_check": {"info": res[0], "iter": res[1], "relres": res[2], "iters_timed": k, "path": pcg_path,
                          "loop": pcg_loop, "loop_info": pcg_loop_info, "bytes_per_iter": pcg_moved},
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "traffic_source": traffic_source,
                "traffic_counters": traffic_detail if isinstance(traffic_detail, dict) else None,
                "algorithmic_bytes_per_launch": kbytes_loc, "avg_launch_ms": kern_ms,
                "median_launch_ms": med_ms,
                "csr_model_bytes_per_launch": csr_model_bytes(n_loc, nnz_loc),
                "csr_model_equiv_GBps": csr_model_bytes(n_loc, nnz_loc) / (kern_ms * 1e-3) / 1e9,
                # the same launch priced in SURVEY 8d's CSR bytes: > 1 is possible and says nothing about the memory system
                # -- csr_spmv_w4 reads no column indices; the comparable figure is `csr_literal` below
                "frac_8d_of_timed_kernel": csr_model_bytes(n_loc, nnz_loc) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                "frac_8d_note": "format-compressed kernel priced in CSR bytes: not comparable, never claimed",
                # the other half of BASELINE.json's metric, where the driver keeps it
                "pcg_iters_per_s": 1.0 / pcg_s_per_iter, "pcg_loop": pcg_loop,
                "pcg_launches_per_iter": pcg_loop_info["launches"], "pcg_bytes_per_iter": pcg_moved,
                "pcg_frac_own_bytes": pcg_moved / pcg_s_per_iter / 1e9 / (HBM_PEAK_GBPS * world),
                "note": "achieved = bytes this kernel's format needs (csr_spmv_w4: 8 B per stored-offset slot + "
                        "2 B row mask + x + y; no column indices) / avg launch time of the K timed launches.  traffic "
                        "above the algorithmic bytes (1.11x at 512^3) is ONE extra pass over x served by the Infinity "
                        "Cache, not DRAM: the per-matrix XCD stripe of 128 workgroups lets two XCDs fetch the same x "
                        "plane at the same time and was chosen because it is 2.5 % FASTER than the stripe of 32 that "
                        "reads 1.007x (DESIGN.md section 3, HISTORY.md 3.1c)",
            },
        }
        out["provenance"] = provenance(L)
        if use_dist:
            out["rccl_ranks"] = dist.get_world_size()  # ranks the process group actually has
            out["backend"] = dist.get_backend()
            out["stage"] = a.stage or None
            out["launcher_kind"] = "one torch.distributed rank per GPU"
            tr = "RCCL" if dist.get_backend() == "nccl" else dist.get_backend()
            out["transport"] = {"halo": "%s send/recv (torch.distributed batch_isend_irecv)" % tr,
                                "reductions": "%s all-reduce %s" % (tr, "in stream order" if tr == "RCCL" else
                                                                    "(host-synchronised around every transfer: rehearsal only)")}
            if preflight is not None:
                out["preflight"] = preflight
            if phases is not None:
                out["phases"] = phases
            if parity_mine is not None:
                out["parity_solve"] = parity_mine
        if dry:
            out["dry_run"] = "launcher / driver plumbing over gloo with " + a.test_backend + ": NOT a measurement"
        if a.share_gpu:
            out["dry_run"] = "%d ranks sharing cuda:0 over %s: a rehearsal of the N > 1 path, NOT a measurement" % (
                world, a.backend)
        if kernels is not None:
            out["kernels_same_operator"] = kernels
            # the other reading of "CSR SpMV % of HBM peak" (VERDICT r4 #5 / #8): the kernels that stream the csr_mat's own
            # col_ind + val arrays (csr_spmv_w6, csr_spmv_w2), priced in SURVEY 8d's bytes -- which is what they move
            lit = [kk for kk in kernels if kk["kernel"] in ("csr_spmv_w6", "csr_spmv_w2")]
            if lit:
                best = max(lit, key=lambda kk: kk["csr_model_frac"])
                out["roofline"]["csr_model_frac_of_streaming_kernel"] = best["csr_model_frac"]
                out["roofline"]["streaming_kernel"] = best["kernel"]
                # the kernel that literally streams the csr_mat's int32 col + fp64 val: SURVEY 8d's bytes ARE its bytes
                out["roofline"]["csr_literal"] = {"kernel": best["kernel"], "avg_launch_ms": best["avg_launch_ms"],
                                                  "frac_8d": best["csr_model_frac"]}
        if sss is not None:
            out["sss_mat"] = sss
        if solvers is not None:
            out["solvers"] = solvers
        if single_kernel is not None:
            out["single_kernel_loops"] = single_kernel
        if strong_n1 is not None:
            out["strong_n1"] = strong_n1
            if strong_n1.get("pcg_iters_per_s"):
                out["roofline"]["strong_n1_iters_per_s"] = strong_n1["pcg_iters_per_s"]
                out["roofline"]["strong_n1_spmv_frac"] = strong_n1.get("spmv_frac_of_peak")
            if use_dist and scaling == "strong" and strong_n1["grid"] == [nx, ny, nz]:
                if strong_n1.get("pcg_iters_per_s"):
                    out["vs_n1"] = (1.0 / pcg_s_per_iter) / strong_n1["pcg_iters_per_s"]
                if parity_mine is not None and strong_n1.get("parity_solve"):
                    par = parity_object(strong_n1["parity_solve"], parity_mine, "n_ranks")
                    out["parity_vs_n1"] = par
                    if not par["ok"]:
                        out["error"] = "parity_vs_n1 failed: max relative difference %.3e > %.1e" % (
                            par["max_rel_diff"], PARITY_TOL)
                        exit_code = 1
        if placement is not None:
            if "first_allocation_ms" in placement:
                placement["first_allocation_ms"] = kern_ms
            out["placement_sweep"] = placement
        if ceiling is not None:
            out["device_ceiling_same_run"] = ceiling
            probe = ceiling.get("read7_write1_probe")
            if probe:
                # the ceiling = the best rate any of this job's plain streaming kernels reached (the 7-read + 1-write
                # probe has the SpMV's shape but csr_spmv_w4 with the round-3 XCD stripe is faster than it; the read-only
                # dot is the fastest of them); a kernel that beats all three is reported at 1.0, not above
                rates = {k: v["GBps"] for k, v in ceiling.items() if isinstance(v, dict) and "GBps" in v}
                best = max(rates, key=rates.get)
                out["roofline"]["stream_ceiling_GBps"] = rates[best]
                out["roofline"]["stream_ceiling_kernel"] = best
                out["roofline"]["frac_of_stream_ceiling"] = min(1.0, achieved / rates[best])
                out["roofline"]["vs_read7_write1_probe"] = achieved / probe["GBps"]
                # which timing level this job's allocations landed on (profiles/r4_modes.txt): named from the
                # dominant kernel's own median launch (512^3 csr_spmv_w4 at the stripe-128 default: <= 1.56 ms fast, >= 1.63 ms slow, 1.60-1.62 usual; profiles/r3_w4_stripe.txt); the counter
                # that moves with it -- read requests the L2s keep in flight, at an unchanged latency per request --
                # comes from this job's profiled child process (`counters`), which may sit in the other mode
                mc = traffic_detail.get("mode_counters") if isinstance(traffic_detail, dict) else None
                if kernel == "csr_spmv_w4" and (nx, ny, nz) == (512, 512, 512):
                    cls = "fast" if med_ms <= 1.56 else ("slow" if med_ms >= 1.63 else "usual")
                else:
                    cls = None
                out["roofline"]["placement_level"] = cls
                out["process_mode"] = {"class": cls, "median_launch_ms": med_ms, "read7_write1_GBps": probe["GBps"],
                                       "counters": mc,
                                       "note": "which LEVEL this job's allocations landed on, not a property of the process: "
                                               "round 4 (profiles/r4_modes.txt) re-allocated nothing but y, or nothing but x, "
                                               "inside one process and moved the same launch across the whole 1.52-1.70 ms range; "
                                               "the byte offset inside an allocation and the memory type do not matter, the "
                                               "dispatcher is ruled out (a persistent grid with a software-defined XCD order does "
                                               "not follow it).  Placement in device memory decides; user space cannot choose it"}
        if clocks is not None:
            out["gpu_clocks_under_load"] = clocks
        if world == 1 and not a.no_cpu_baseline:
            base, ref, parity = cpu_baseline(dev=dev, published=not a.grid)
            if "published_table" in base:
                out["published_table"] = base.pop("published_table")
            if parity is not None:
                out["parity_check"] = parity
                if not parity["ok"]:
                    out["error"] = "parity_check failed (GPU iterates against the oracle at BASELINE's sizes)"
                    exit_code = 1
            out["cpu_baseline"] = base
            out["cpu_baseline"]["gpu_over_cpu"] = out["effective_csr_model_GBps"] / base["value"]
            # configs[0] (poisson2d(100)) on the GPU beside the CPU's C1 figure: the single-kernel loop of psp_coop.hip
            try:
                A1 = dev.DeviceCSR.poisson(100, 100)
                K1 = dev.DeviceJacobi(A1)
                b1 = np.empty(10000)
                A1.matvec(np.ones(10000), b1)
                dev.pcg(A1, b1, np.zeros(10000), 0.0, 50, K1)
                best, its = 1e9, 1
                for _ in range(3):
                    t = time.perf_counter()
                    r1 = dev.pcg(A1, b1, np.zeros(10000), 0.0, 2000, K1)
                    dt = time.perf_counter() - t
                    if dt < best:
                        best, its = dt, min(r1[1], 2000) + 1
                out["cpu_baseline"]["C1_poisson2d_100"]["gpu_pcg_iters_per_s"] = its / best
                out["cpu_baseline"]["C1_poisson2d_100"]["gpu_pcg_us_per_iter"] = best / its * 1e6
                out["cpu_baseline"]["C1_poisson2d_100"]["gpu_iterations_run"] = its - 1
            except Exception as e:  # noqa: BLE001 - a reported extra, never fatal for the bench line
                out["cpu_baseline"]["C1_poisson2d_100"]["gpu_error"] = str(e)[:200]
            # configs[1] (4096^2) on the GPU beside the CPU's C2 figure: product and Jacobi-PCG, device-resident vectors
            try:
                c2key = [k for k in base if k.startswith("C2_")][0]
                g2 = int(c2key.rsplit("_", 1)[1])
                A2 = dev.DeviceCSR.poisson(g2, g2)
                n2 = A2.shape[0]
                x2 = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n2))
                y2 = dev.DeviceBuffer(n2)
                f2 = lambda: A2.matvec_dev(x2.ptr, y2.ptr)  # noqa: E731
                timed_launches(f2, sync, ev, 5)
                base[c2key]["gpu_spmv_ms"] = timed_launches(f2, sync, ev, 20)[0]
                x2.free()
                y2.free()
                t2, res2 = pcg_single(L, check, dev, A2, n2, 200, sync)
                base[c2key]["gpu_pcg_iters_per_s"] = 1.0 / t2
                base[c2key]["gpu_pcg_check"] = {"info": res2[0], "iter": res2[1]}
                del A2
            except Exception as e:  # noqa: BLE001
                base[[k for k in base if k.startswith("C2_")][0]]["gpu_error"] = str(e)[:200]
            if ref is not None:
                out["cpu_baseline_reference_pcg"] = ref
            # SURVEY 8d's optional second line, labelled: NOT the reference (which is one thread) -- the headline case's rows
            # over every host core this job may use; never the baseline, it only places the one-core figure
            heads = [v for k, v in base.items() if isinstance(v, dict) and k.startswith(("C3_", "poisson3d_"))]
            mt = heads[0].get("all_cores_non_reference") if heads else None
            if mt:
                out["cpu_all_cores_non_reference"] = {
                    "value": mt["spmv_GBps"], "unit": "GB/s", "cores": mt["threads"], "kind": "port",
                    "sample": "the same CSR SpMV (n=%d), rows in contiguous ranges over %d POSIX threads, median of >= 5"
                              % (heads[0]["n"], mt["threads"]),
                    "note": "non-reference: pysparse's product is single-threaded (no OpenMP, GIL held); every y[i] has the "
                            "one-thread bits (%s)" % mt["same_bits_as_one_thread"],
                    "gpu_over_all_cores": out["effective_csr_model_GBps"] / mt["spmv_GBps"]}
        if world == 1 and not use_dist and os.environ.get("EMILIA_MTX"):
            try:  # configs[4] on the user's file, beside the headline (its own failure never costs the line)
                out["config5"] = mtx_leg(os.environ["EMILIA_MTX"])
            except Exception as e:  # noqa: BLE001
                out["config5"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        if use_dist and world > 1:
            # DESIGN.md section 5's prediction for THIS N beside what was measured
            t1 = (1e3 / strong_n1["pcg_iters_per_s"]) if (strong_n1 and strong_n1.get("pcg_iters_per_s")
                                                          and strong_n1["grid"] == [nx, ny, nz]) else None
            out["predicted"] = predicted_iteration(n_tot, world, nx * ny, t1, kbytes_row=kbytes_loc / float(n_loc))
            out["predicted"]["missed_budget"] = judge_phases(phases, out["predicted"])
        emit(out, real_stdout, a.side_file or None)
    if use_dist:
        code = torch.tensor([exit_code], dtype=torch.int32, device="cpu" if dry else "cuda")
        dist.all_reduce(code, op=dist.ReduceOp.MAX)  # every rank leaves with rank 0's verdict
        exit_code = int(code.item())
        dist.barrier()
        dist.destroy_process_group()
    return exit_code


if __name__ == "__main__":
    raise SystemExit(main())
