"""bench_legs.py -- the legs beside bench.py's timed region (split out of bench.py in round 5): hardware-counter child runs
for `roofline.traffic`, clocks, the other SpMV kernels on the same operator, the placement sweep, the streaming ceilings,
the sss_mat leg, the one-GPU 1024^3 leg, the single-process N-GPU driver.  They REPORT beside `value`; none of them feeds
it.  Nothing here imports the oracle."""
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

from bench_common import (HBM_PEAK_GBPS, METRIC, PARITY_ITERS, PARITY_TOL, ROOT, W2_VARIANT, W3_VARIANT, W6_VARIANT,  # noqa: F401
                          Events, csr_model_bytes, kernel_bytes, parity_object, provenance, timed_launches)

def live_traffic(grid, variant):
    """HBM-side bytes per launch of the SpMV kernel(s) of this operator, measured in THIS job: two child processes
    under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md, HBM section: on
    gfx950 FETCH_SIZE reports half the bytes of a wide streaming read -> doubled; both in KB) run the same operator
    through tools/prof_spmv.py.  Called BEFORE this process touches the GPU: with a second process holding a context
    on the device a counter pass takes minutes instead of seconds.  Returns ({kernel name: {...}}, None) or
    (None, reason): no profiler, a profiler already attached to this process, a time-out -- the caller then falls
    back to the committed passes."""
    import csv
    import glob
    import shutil
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or \
            "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process runs under a profiler"
    tmp = tempfile.mkdtemp(prefix="psp_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [prof, "--pmc", ctr, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "tools", "prof_spmv.py"), "--reps", "3", "--grid", "%d,%d,%d" % grid,
                   "--variant", str(variant)]
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=90)
            if r.returncode != 0:
                return None, "rocprofv3 --pmc %s exited with %d" % (ctr, r.returncode)
            acc = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    name = row.get("Kernel_Name", "")
                    if "_spmv_" in name and row.get("Counter_Name") == ctr:
                        acc.setdefault(name, []).append(float(row["Counter_Value"]))
            if not acc:
                return None, "no %s samples of an SpMV kernel" % ctr
            for name, v in acc.items():
                vals.setdefault(name, {})[ctr] = sum(v) / len(v)
        # third pass (round 3, profiles/r3_modes.txt): what differs between a fast and a slow process of the same launch
        # is not bytes, clocks or the latency of a memory request but HOW MANY read requests the L2s keep in flight:
        # TCC_EA0_RDREQ_LEVEL / TCC_CYCLE (reads in flight, summed over the channels) and RDREQ_LEVEL / RDREQ (cycles per
        # request), with the kernel's duration in that process.  Best effort: a failure only drops the field.
        try:
            grp = ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum", "TCC_CYCLE_sum", "GRBM_GUI_ACTIVE"]
            outd = os.path.join(tmp, "mode")
            cmd = [prof, "--pmc"] + grp + ["--output-format", "csv", "-d", outd, "--", sys.executable,
                                           os.path.join(ROOT, "tools", "prof_spmv.py"), "--reps", "5", "--grid",
                                           "%d,%d,%d" % grid, "--variant", str(variant)]
            r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=90)
            if r.returncode == 0:
                acc, dur = {}, {}
                for f in glob.glob(os.path.join(outd, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        name = row.get("Kernel_Name", "")
                        if "_spmv_" in name:
                            acc.setdefault(name, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                            dur.setdefault(name, {})[row.get("Dispatch_Id")] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
                for name, c in acc.items():
                    m = {k: sum(v) / len(v) for k, v in c.items()}
                    d = sorted(dur[name].values())
                    if m.get("TCC_EA0_RDREQ_sum") and m.get("TCC_CYCLE_sum"):
                        vals.setdefault(name, {})["mode"] = {
                            "kernel_ms_in_that_process": d[len(d) // 2],
                            # both are sums over the 128 channel instances: their ratio is the average per channel
                            "ea_reads_in_flight_per_channel": m["TCC_EA0_RDREQ_LEVEL_sum"] / m["TCC_CYCLE_sum"],
                            "ea_read_latency_tcc_cycles": m["TCC_EA0_RDREQ_LEVEL_sum"] / m["TCC_EA0_RDREQ_sum"],
                            "gpu_cycles": m.get("GRBM_GUI_ACTIVE"),
                        }
        except (OSError, subprocess.SubprocessError, ValueError, KeyError):
            pass
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return None, "%s: %s" % (type(e).__name__, e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for name, v in vals.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            out[name] = {"bytes": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0, "FETCH_SIZE_KB": v["FETCH_SIZE"],
                         "WRITE_SIZE_KB": v["WRITE_SIZE"], "fetch_correction": 2.0}
            if "mode" in v:
                out[name]["mode_counters"] = v["mode"]
    return (out, None) if out else (None, "counters incomplete")


def gpu_clocks():
    """rocm-smi, called while ~1 s of SpMV launches is in flight: which clock / power state the numbers
    of this run come from (runs land in a faster and a slower mode per box)."""
    try:
        p = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showperflevel", "--showtemp",
                            "--json"],
                           capture_output=True, text=True, timeout=20)
        txt = p.stdout.strip()
        try:
            d = json.loads(txt)
            card = d.get("card0", d)
            keep = {}
            for k, v in card.items():
                kl = k.lower()
                if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "power", "performance level", "temperature")):
                    keep[k] = v
            return keep or {"raw": txt[:400]}
        except ValueError:
            return {"raw": txt[:400] or p.stderr.strip()[:400]}
    except (OSError, subprocess.SubprocessError) as e:
        return {"error": str(e)[:200]}


def peer_matrix(L, ndev):
    """pre-flight: can device i reach device j's memory directly (psp_peer_access; no context is created)"""
    m = []
    for i in range(ndev):
        row = []
        for j in range(ndev):
            c = C.c_int(-1)
            row.append(c.value if L.psp_peer_access(i, j, C.byref(c)) == 0 else -1)
        m.append(row)
    return m


def link_topology():
    """pre-flight: how the GPUs are wired (rocm-smi --showtopotype: XGMI / PCIE per pair), best effort"""
    try:
        p = subprocess.run(["rocm-smi", "--showtopotype", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(p.stdout)
    except (OSError, subprocess.SubprocessError, ValueError) as e:
        return {"error": str(e)[:200]}


def pcg_single(L, check, dev, A, n, iters, sync, parity=False):
    """Jacobi-PCG through the library's device-resident loop: b = A*ones, x0 = 0, tol = 0 (exactly
    `iters` iterations; ||b|| and r = b - A x0 are inside the timed region).  parity: the warm-up solve runs
    PARITY_ITERS iterations and leaves (relres, x.b, x.x) as the third result."""
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    ones = np.ones(1 << 24)
    for k in range(0, n, ones.size):  # chunked: n may be 2^30
        check(L.psp_memcpy_h2d(xb.ptr + 8 * k, ones.ctypes.data, 8 * min(ones.size, n - k)))
    A.matvec_dev(xb.ptr, bb.ptr)
    sync()
    par = None
    for kk in ((PARITY_ITERS if parity else 2), iters):  # first call = warm-up
        xb.zero()
        info, it, rr = C.c_int(), C.c_int(), C.c_double()
        sync()
        t = time.perf_counter()
        check(L.psp_pcg_dev(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it),
                            C.byref(rr), None))
        sync()
        dt = time.perf_counter() - t
        if parity and par is None:
            ob = dev.DeviceBuffer(2)
            check(L.psp_k_dot(n, xb.ptr, bb.ptr, ob.ptr))
            check(L.psp_k_dot(n, xb.ptr, xb.ptr, ob.ptr + 8))
            sync()
            v = ob.download()
            ob.free()
            par = {"relres": rr.value, "x_dot_b": float(v[0]), "x_dot_x": float(v[1]),
                   "info_iter": [info.value, it.value]}
    del aop, kop, K
    bb.free()
    xb.free()
    if parity:
        return dt / iters, (info.value, it.value, rr.value), par
    return dt / iters, (info.value, it.value, rr.value)


def strong_n1_leg(L, check, dev, grid, iters):
    """the whole strong-scaling problem on ONE GPU (index-free operator, psp_csr_poisson_big): SpMV
    time and Jacobi-PCG iterations/s -- the denominator of `vs_n1`"""
    def sync():
        check(L.psp_synchronize())
    nx, ny, nz = grid
    A = dev.DeviceCSR.poisson_big(nx, ny, nz)
    n, nnz = A.shape[0], A.nnz
    x, y = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    chunk = np.random.default_rng(0).standard_normal(1 << 24)
    for k in range(0, n, chunk.size):
        check(L.psp_memcpy_h2d(x.ptr + 8 * k, chunk.ctypes.data, 8 * min(chunk.size, n - k)))
    ev = Events(L, check, 12)
    timed_launches(lambda: A.matvec_dev(x.ptr, y.ptr), sync, ev, 3)
    avg, med = timed_launches(lambda: A.matvec_dev(x.ptr, y.ptr), sync, ev, 10)
    kern, info = A.kernel_info()
    x.free()
    y.free()
    s_per_it, chk, par = pcg_single(L, check, dev, A, n, iters, sync, parity=True)
    kb = kernel_bytes(kern, info, n, nnz)
    out = {"grid": [nx, ny, nz], "n": n, "nnz": nnz, "kernel": kern, "spmv_ms": med,
           "spmv_GBps": kb / (med * 1e-3) / 1e9, "spmv_frac_of_peak": kb / (med * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "pcg_iters_per_s": 1.0 / s_per_it, "pcg_check": {"info": chk[0], "iter": chk[1], "relres": chk[2]},
           "parity_solve": par,  # after PARITY_ITERS iterations: what an N-rank solve of the same system is held to
           "path": "psp_pcg_dev (single-GPU device-resident loop)"}
    A.close()
    check(L.psp_trim())
    return out


def dry_strong_n1(test_backend, grid):
    """CPU dry run of the launcher (tests): the parity reference = the same system solved by the test backend without
    a partition (SingleComm), PARITY_ITERS iterations"""
    import importlib
    from pysparse_amd import distributed as D
    mod, fn = test_backend.split(":")
    be, make_local = getattr(importlib.import_module(mod), fn)()
    nx, ny, nz = grid
    A = D.DistCSR.poisson(nx, ny, nz, D.SingleComm(), be, make_local)
    n = A.n_local
    ones = A.new_ext()
    ones.fill_(1.0)
    b = be.zeros(n)
    A.matvec(ones, b)
    dinv = be.zeros(n)
    dinv.fill_(1.0 / (6.0 if nz > 0 else 4.0))
    x = be.zeros(n)
    res = D.dist_pcg(A, b, x, 0.0, PARITY_ITERS, dinv)
    return {"grid": [nx, ny, nz], "n": n, "path": "test backend, world size 1",
            "parity_solve": {"relres": res[2], "x_dot_b": float(be.dot(x, b)[0]), "x_dot_x": float(be.dot(x, x)[0]),
                             "info_iter": [res[0], res[1]]}}


def single_process_main(a):
    """--single-process: the N-GPU job as ONE process through the C ABI's device-list variant
    (psp_csr_poisson_multi: one rank per entry of the list, peer copies for the ghost planes, RCCL for the two
    packed reductions of an iteration) -- what `krylov.pcg(A, ...)` of a drop-in script runs when A was made with
    devices=[...].  Same workload and the same JSON line as the torch.distributed launch (strong scaling at 1024^3
    for N > 1).  --share-gpu lists device 0 N times: a rehearsal on a one-GPU box, not a measurement."""
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    from pysparse_amd import _capi, device as dev
    L, check = _capi.lib(), _capi.check
    N = a.gpus
    devices = [0] * N if a.share_gpu else list(range(N))
    if a.grid:
        nx, ny, nz = (int(t) for t in a.grid.split(","))
    else:
        nx = ny = nz = 1024 if N > 1 else 512
    preflight = {"device_count": L.psp_device_count(), "peer_access": peer_matrix(L, 1 if a.share_gpu else N)}
    if N > 1 and not a.share_gpu:
        preflight["link_topology"] = link_topology()
    # the 1-GPU end of the strong-scaling ratio and of the parity check: the whole problem on device 0, first
    strong_n1 = None
    if N > 1 and not a.no_strong_n1:
        check(L.psp_set_device(0))
        strong_n1 = strong_n1_leg(L, check, dev, (nx, ny, nz), min(a.pcg_iters, 24))
    A = dev.DeviceCSR.poisson_multi(nx, ny, nz, devices=devices)
    n, nnz = A.shape[0], A.nnz
    ranks, distinct, rccl_used = A.multi_info()
    ms = C.c_double()
    t0 = time.perf_counter()
    check(L.psp_csr_multi_spmv_time(A._h, a.warmup, a.steps, C.byref(ms)))
    wall = time.perf_counter() - t0
    kbytes = 8 * 7 * n + 2 * n + 16 * n  # csr_spmv_w4 on the 7-point operator (kernel_bytes)
    if nz == 0:
        kbytes = 8 * 5 * n + 2 * n + 16 * n
    # the pieces of that product and of an iteration on their own (psp_csr_multi_phase_time)
    phases = {}
    for what, name in ((0, "halo_ms"), (1, "spmv_local_ms"), (2, "allreduce_ms")):
        v = C.c_double()
        check(L.psp_csr_multi_phase_time(A._h, what, 2, max(5, min(a.steps, 20)), C.byref(v)))
        phases[name] = v.value
    phases["spmv_with_halo_ms"] = ms.value
    phases["allreduce_us"] = phases["allreduce_ms"] * 1e3
    if phases["halo_ms"] > 0:
        phases["overlap_frac"] = max(0.0, min(1.0, (phases["halo_ms"] + phases["spmv_local_ms"] - ms.value)
                                              / phases["halo_ms"]))
    # Jacobi-PCG iterations/s: tol = 0 runs exactly k iterations; two runs, the difference cancels the host
    # transfers of b and x and the set-up products (the vectors cross PCIe once per solve)
    K = dev.DeviceJacobi(A)
    ones = np.ones(n)
    b = np.empty(n)
    A.matvec(ones, b)
    del ones
    # in-job parity: PARITY_ITERS iterations against the one-GPU solve of the same system (strong_n1)
    parity = None
    x = np.zeros(n)
    rp = dev.pcg(A, b, x, 0.0, PARITY_ITERS, K)
    mine = {"relres": rp[2], "x_dot_b": float(np.dot(x, b)), "x_dot_x": float(np.dot(x, x)), "info_iter": [rp[0], rp[1]]}
    if strong_n1 is not None:
        parity = parity_object(strong_n1["parity_solve"], mine, "n_ranks")
    k1, k2 = 4, 4 + max(8, min(a.pcg_iters, 64))
    # the difference of two solves of k1 and k2 iterations (best of three each, after a warm-up solve): on a problem
    # of a few hundred thousand rows the host-side noise of a single pair can exceed the k2 - k1 iterations themselves
    times = {}
    for k in (k1, k1, k2, k1, k2, k1, k2):
        x = np.zeros(n)
        t = time.perf_counter()
        res = dev.pcg(A, b, x, 0.0, k, K)
        dt = time.perf_counter() - t
        times[k] = min(times.get(k, dt), dt) if k in times or k != k1 else dt
    if times[k2] > times[k1]:
        s_per_iter = (times[k2] - times[k1]) / (k2 - k1)
    else:  # still inside the noise: price the whole longer solve (an upper bound of the iteration time)
        s_per_iter = times[k2] / k2
    reductions = "rccl" if rccl_used else ("none" if ranks == 1 else "fold kernel over peer pointers")
    out = {
        "metric": METRIC,
        "value": kbytes / (ms.value * 1e-3) / 1e9, "unit": "GB/s", "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": ms.value, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "3D Poisson 7-pt %dx%dx%d fp64 csr_mat, y = A x; z-slab row partition over a device list, "
                               "ONE process (psp_csr_poisson_multi)" % (nx, ny, nz),
                   "n": n, "nnz": nnz, "rows_per_gpu": n // N, "parallelism": "row-range x%d, single process" % N,
                   "scaling_mode": "strong", "devices": devices},
        "launcher_kind": "single process, C ABI device list",
        "stage": a.stage or None,
        "transport": {"halo": "hipMemcpyPeerAsync on a copy stream per rank" if distinct > 1 else
                              "device-to-device copies (ranks share a GPU)",
                      "reductions": reductions},
        "ranks": ranks, "distinct_devices": distinct, "rccl_ranks": ranks if rccl_used else 0,
        "reductions": reductions,
        "pct_hbm_peak": 100.0 * kbytes / (ms.value * 1e-3) / 1e9 / (HBM_PEAK_GBPS * max(distinct, 1)),
        "pcg_iters_per_s": 1.0 / s_per_iter,
        "pcg_check": {"info": res[0], "iter": res[1], "relres": res[2], "iters_timed": k2 - k1,
                      "path": "psp_pcg on a multi-device matrix (psp_multi.hip)", "solve_s": times},
        "phases": phases,
        "preflight": preflight,
        "parity_solve": mine,
        "provenance": provenance(L),
        "wall_s_spmv_leg": wall,
    }
    if strong_n1 is not None:
        out["strong_n1"] = strong_n1
        out["vs_n1"] = (1.0 / s_per_iter) / strong_n1["pcg_iters_per_s"]
    if parity is not None:
        out["parity_vs_n1"] = parity
        if not parity["ok"]:
            out["error"] = "parity_vs_n1 failed: max relative difference %.3e > %.1e" % (parity["max_rel_diff"], PARITY_TOL)
    if a.share_gpu:
        out["dry_run"] = "%d ranks sharing device 0 in one process: a rehearsal of the N > 1 path, NOT a measurement" % N
    out["roofline"] = {"bound": "hbm", "kernel": "csr_spmv_w4", "achieved": out["value"] / max(distinct, 1), "peak": HBM_PEAK_GBPS,
                       "unit": "GB/s", "frac": out["value"] / max(distinct, 1) / HBM_PEAK_GBPS, "traffic": None,
                       "algorithmic_bytes_per_launch": kbytes // N, "avg_launch_ms": ms.value,
                       "pcg_iters_per_s": 1.0 / s_per_iter, "pcg_loop": "pcg_multi",
                       "pcg_bytes_per_iter": kbytes + 64 * n,
                       "pcg_frac_own_bytes": (kbytes + 64 * n) / s_per_iter / 1e9 / (HBM_PEAK_GBPS * max(distinct, 1))}
    if strong_n1 is not None:
        out["roofline"]["strong_n1_iters_per_s"] = strong_n1["pcg_iters_per_s"]
    if N > 1:
        from bench_line import judge_phases, predicted_iteration
        t1 = 1e3 / strong_n1["pcg_iters_per_s"] if strong_n1 is not None else None
        out["predicted"] = predicted_iteration(n, N, nx * ny, t1, kbytes_row=kbytes / float(n))
        out["predicted"]["missed_budget"] = judge_phases(dict(phases, iteration_ms=s_per_iter * 1e3), out["predicted"])
    from bench_line import emit
    emit(out, real_stdout, getattr(a, "side_file", "") or None)
    return 1 if "error" in out else 0


# ------------------------------------------------------------------------------------ legs beside the timed region (N = 1)
# ctx: what bench.py's body holds for its operator -- L, check, dev, A, xb, yb, n, nnz, step, sync, ev, steps, kbytes

def same_operator_kernels_leg(ctx):
    """`kernels_same_operator`: the other SpMV kernels on the SAME operator and vectors.  csr_spmv_w4 (the default for a
    stencil operator) reads no column indices; csr_spmv_w3 is what an arbitrary banded csr_mat gets (16-bit chunk-local
    columns); csr_spmv_w6 (round 5) and csr_spmv_w2 stream int32 col + fp64 val exactly as the csr_mat stores them."""
    A, step, sync, ev, n, nnz = ctx["A"], ctx["step"], ctx["sync"], ctx["ev"], ctx["n"], ctx["nnz"]
    kernels = []
    for var in (W3_VARIANT, W6_VARIANT, W2_VARIANT):
        A.set_variant(var)
        kn, ki = A.kernel_info()
        timed_launches(step, sync, ev, 3)
        avg, med = timed_launches(step, sync, ev, ctx["steps"])
        own = kernel_bytes(kn, ki, n, nnz)
        kernels.append({"kernel": kn, "avg_launch_ms": avg, "median_launch_ms": med,
                        "bytes_per_launch": own, "GBps": own / (avg * 1e-3) / 1e9,
                        "frac": own / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "csr_model_GBps": csr_model_bytes(n, nnz) / (avg * 1e-3) / 1e9,
                        "csr_model_frac": csr_model_bytes(n, nnz) / (avg * 1e-3) / 1e9 / HBM_PEAK_GBPS})
    A.set_variant(-1)
    return kernels


def placement_sweep_leg(ctx):
    """`placement_sweep`: the placement levels (profiles/r4_modes.txt).  What a launch takes depends on
    where in device memory its operands lie; `value` is THIS job's first allocation, as any job's would be.  Here y is
    re-allocated six times (pads of odd sizes in between, everything stays alive until the end) and the same launch is
    timed on each: the spread a user sees, in every bench line.  Reported only."""
    dev, A, xb, sync, ev, n = ctx["dev"], ctx["A"], ctx["xb"], ctx["sync"], ctx["ev"], ctx["n"]
    keep, ms_list = [], []
    try:
        for j in range(6):
            keep.append(dev.DeviceBuffer((37 + 101 * j) * (1 << 17) + 512 * j))  # (37 + 101 j) MiB + 4 j KiB
            yj = dev.DeviceBuffer(n)
            keep.append(yj)
            fj = lambda yj=yj: A.matvec_dev(xb.ptr, yj.ptr)  # noqa: E731
            timed_launches(fj, sync, ev, 3)
            ms_list.append(timed_launches(fj, sync, ev, min(ctx["steps"], 20))[0])
        placement = {"what": "the same launch with y re-allocated six times (x and the operator stay): where the operands lie "
                             "decides up to 8 % (profiles/r4_modes.txt); `value` is the job's FIRST allocation",
                     "y_realloc_avg_launch_ms": ms_list, "first_allocation_ms": None,
                     "best_ms": min(ms_list), "worst_ms": max(ms_list),
                     "best_frac_of_peak": ctx["kbytes"] / (min(ms_list) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "worst_frac_of_peak": ctx["kbytes"] / (max(ms_list) * 1e-3) / 1e9 / HBM_PEAK_GBPS}
    except Exception as e:  # noqa: BLE001 - a reported extra (e.g. out of memory on a small device), never fatal
        placement = {"error": str(e)[:200]}
    for bfr in keep:
        bfr.free()
    return placement


def stream_ceiling_leg(ctx):
    """`device_ceiling_same_run`: what this GPU's memory system gives the library's own streaming kernels in the same run
    ("a measured device ceiling from the same run") -- a read-only pass (the dot product kernel: 16 n bytes), a
    read-read-write pass (y = x o dinv: 24 n bytes) over the same vectors, and the access shape of the dominant kernel as
    a plain streaming kernel: 7 read streams + 1 write stream of 1 GiB each (psp_stream_probe; csr_spmv_w4 on the
    7-point operator reads 7 value streams and writes y)."""
    L, check, dev, xb, yb, sync, ev, n = (ctx[k] for k in ("L", "check", "dev", "xb", "yb", "sync", "ev", "n"))
    zb = dev.DeviceBuffer(n)
    ob = dev.DeviceBuffer(1)
    check(L.psp_k_jacobi(n, xb.ptr, xb.ptr, zb.ptr))  # fill zb

    def dot_step():
        check(L.psp_k_dot(n, xb.ptr, zb.ptr, ob.ptr))

    def triad_step():
        check(L.psp_k_jacobi(n, xb.ptr, zb.ptr, yb.ptr))
    ceiling = {"what": "library streaming kernels on vectors of n = %d fp64, same process" % n}
    for name, fn, nbytes in (("read_only_dot", dot_step, 16 * n), ("read2_write1", triad_step, 24 * n)):
        timed_launches(fn, sync, ev, 3)
        avg, med = timed_launches(fn, sync, ev, min(ctx["steps"], 50))
        ceiling[name] = {"bytes": nbytes, "avg_launch_ms": avg, "GBps": nbytes / (avg * 1e-3) / 1e9}
    zb.free()
    ob.free()
    pa, pm = C.c_float(), C.c_float()
    check(L.psp_stream_probe(7, 1, 1 << 30, 10, C.byref(pa), C.byref(pm)))
    ceiling["read7_write1_probe"] = {"bytes": 8 << 30, "avg_launch_ms": pa.value, "min_launch_ms": pm.value,
                                     "GBps": (8 << 30) / (pa.value * 1e-3) / 1e9}
    return ceiling


def sss_leg(ctx):
    """`sss_mat`: the same operator as an sss_mat (examples/poisson_test.py solves with S = L.to_sss()): y = S x from the
    strict lower triangle only, and Jacobi-PCG on it"""
    L, check, dev, xb, yb, sync, ev, n = (ctx[k] for k in ("L", "check", "dev", "xb", "yb", "sync", "ev", "n"))
    S = dev.DeviceSSS.poisson(*ctx["grid"])

    def sstep():
        S.matvec_dev(xb.ptr, yb.ptr)
    timed_launches(sstep, sync, ev, 3)
    s_avg, s_med = timed_launches(sstep, sync, ev, ctx["steps"])
    nnz_lower = S.nnz - n
    s_per_it, s_chk = pcg_single(L, check, dev, S, n, ctx["pcg_iters"], sync)
    skern, sinfo = S.kernel_info()
    sown = kernel_bytes(skern, sinfo, n, 2 * nnz_lower + n, nnz_lower)
    out = {"kernel": skern, "spmv_ms": s_avg, "median_launch_ms": s_med,
           # B_sss = 12 nnz_lower + 28 n + 4; the kernel's own format moves `bytes_per_launch`
           "bytes_per_launch": sown, "spmv_GBps": sown / (s_avg * 1e-3) / 1e9,
           "frac": sown / (s_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS,
           "sss_model_GBps": (12 * nnz_lower + 28 * n + 4) / (s_avg * 1e-3) / 1e9,
           "pcg_iters_per_s": 1.0 / s_per_it,
           "pcg_check": {"info": s_chk[0], "iter": s_chk[1], "relres": s_chk[2]}}
    S.close()
    return out



# vector streams (8 n bytes each) of one iteration of the fused loops (psp_solvers.hip: cgs_device / bicgstab_device /
# qmrs_device, native operator + jacobi(1) with a constant diagonal: dinv is a scalar, not a stream) and the products in it:
#   cgs       v = A K p (+ v.r0): 3 | q, tmp2, x: 6 | t = A tmp2: 2 | r, {r.r, r.r0}: 4 | u, p, K p: 6
#   bicgstab  p, K p: 5 | v = A phat (+ rhat.v): 3 | s, K s: 4 | t = A shat (+ t.s): 3 | t.t: 1 | x, r, {r.r, rhat.r}: 8
#   qmrs      p, g: 6 | t = A g (+ g.t): 2 | v1, v1.v1: 3 | d, x, v1, K v1: 8
#   gmres(m)  per inner iteration i: W_i = K V_i: 2 | V_{i+1} = A W_i: 2 | V_{i+1}.V_0: 2 | i + 1 modified Gram-Schmidt steps of
#             4 streams (the last: 3) | scaling: 2   = 4 i + 11; per restart: r = b - A x: 2 + 3, its norm and scaling: 1 + 2,
#             x += s_j W_j: 3 m
SOLVER_STREAMS = {"cgs": (2, 21), "bicgstab": (2, 24), "qmrs": (1, 19)}


def gmres_streams(m):
    """(products, vector streams) per inner iteration of gmres(m), averaged over whole restart cycles"""
    inner = sum(4 * i + 11 for i in range(m))
    restart = 2 + 3 + 1 + 2 + 3 * m
    return 1.0 + 1.0 / m, (inner + restart) / m


def solvers_leg(ctx):
    """`solvers` (VERDICT r4 'Next' #5): the reference's other Krylov solvers (pysparse/itsolvers/src/{cgs,bicgstab,qmrs,
    gmres}.c) on the same operator with Jacobi, through the host-pointer entry points the extension module binds; two
    truncated solves of different length (tol = 0) per solver, so that the copies of b and x and the set-up drop out.
    Priced in the bytes their kernels have to move: the operator once per product + the vector streams above."""
    dev, A, n, kbytes = (ctx[k] for k in ("dev", "A", "n", "kbytes"))
    K = dev.DeviceJacobi(A)
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    m = 20
    op_bytes = kbytes - 16 * n  # the operator's own bytes of one product (kbytes counts x and y once)
    out = {"what": "Jacobi-preconditioned, b = A*ones, x0 = 0, tol = 0; per-iteration time from two truncated solves; "
                   "bytes = products x operator bytes + vector streams x 8 n (bench_legs.SOLVER_STREAMS)",
           "operator_bytes_per_product": op_bytes}
    for name, fn, short, long_ in (("cgs", dev.cgs, 5, 25), ("bicgstab", dev.bicgstab, 5, 25), ("qmrs", dev.qmrs, 5, 25),
                                   ("gmres20", dev.gmres, m, 3 * m)):
        ts, res = {}, None
        fn(A, b, np.zeros(n), 0.0, short, K)  # warm-up: the solver's work vectors are allocated (and pooled) here
        for kk in (short, long_, short, long_, short, long_):
            x = np.zeros(n)
            t = time.perf_counter()
            res = fn(A, b, x, 0.0, kk, K)
            ts.setdefault(kk, []).append(time.perf_counter() - t)
        # medians: a single slow or fast sample (host copies of 1 GB vectors are inside both) must not move the difference
        dt = (float(np.median(ts[long_])) - float(np.median(ts[short]))) / (long_ - short)
        prods, streams = gmres_streams(m) if name == "gmres20" else SOLVER_STREAMS[name]
        nbytes = prods * op_bytes + streams * 8 * n
        out[name] = {"ms_per_iter": dt * 1e3, "iters_per_s": 1.0 / dt if dt > 0 else None,
                     "products_per_iter": prods, "vector_streams_per_iter": streams, "bytes_per_iter": nbytes,
                     "GBps": nbytes / dt / 1e9 if dt > 0 else None,
                     "frac": nbytes / dt / 1e9 / HBM_PEAK_GBPS if dt > 0 else None,
                     "last": [int(res[0]), int(res[1]), float(res[2])]}
    del K
    return out


def single_kernel_leg(L, check, dev):
    """`single_kernel_loops`: Jacobi-PCG and Jacobi-MINRES per iteration where the whole loop runs as ONE cooperative kernel
    (pysparse_amd/csrc/psp_coop.hip, psp_mid.hip: row blocks for 2-D grids, bricks for 3-D ones) -- configs[0]'s size, a 2-D
    and a 3-D grid of 10^6 points; two truncated solves each (tol = 0), device-resident vectors.  Reported only."""
    out = {"what": "microseconds per iteration, Jacobi, b = A x_random, x0 = 0, tol = 0 (two truncated solves); which loop ran: "
                   "counters of the library"}
    for grid, (k1, k2) in (((100, 100, 0), (40, 140)), ((1024, 1024, 0), (100, 600)), ((100, 100, 100), (15, 75))):
        A = dev.DeviceCSR.poisson(*grid)
        n = A.shape[0]
        K = dev.DeviceJacobi(A)
        aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
        bb, xb = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
        xb.upload(np.random.default_rng(1).standard_normal(n))
        A.matvec_dev(xb.ptr, bb.ptr)
        check(L.psp_synchronize())
        rec = {"n": n}
        m0, b0 = (C.c_longlong(), C.c_longlong()), (C.c_longlong(), C.c_longlong())
        L.psp_debug_mid_count(C.byref(m0[0]), C.byref(m0[1]))
        L.psp_debug_brick_count(C.byref(b0[0]), C.byref(b0[1]))
        for name, fn in (("pcg", L.psp_pcg_dev), ("minres", L.psp_minres_dev)):
            ts = {}
            for kk in (k1, k1, k2, k1, k2):
                xb.zero()
                info, it, rr = C.c_int(), C.c_int(), C.c_double()
                check(L.psp_synchronize())
                t = time.perf_counter()
                check(fn(aop._h, kop._h, n, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it), C.byref(rr), None))
                check(L.psp_synchronize())
                ts.setdefault(kk, []).append(time.perf_counter() - t)
            rec[name + "_us_per_iter"] = (min(ts[k2]) - min(ts[k1])) / (k2 - k1) * 1e6
            rec[name + "_last"] = [info.value, it.value]
        m1, b1 = (C.c_longlong(), C.c_longlong()), (C.c_longlong(), C.c_longlong())
        L.psp_debug_mid_count(C.byref(m1[0]), C.byref(m1[1]))
        L.psp_debug_brick_count(C.byref(b1[0]), C.byref(b1[1]))
        rec["loop"] = ("row blocks (psp_mid.hip)" if m1[0].value > m0[0].value else
                       "bricks (psp_mid.hip)" if b1[0].value > b0[0].value else "one row per thread (psp_coop.hip) or launch per phase")
        rec["handed_back"] = int(m1[1].value - m0[1].value + b1[1].value - b0[1].value)
        out["x".join(str(v) for v in grid if v)] = rec
        del aop, kop, K
        A.close()
        bb.free()
        xb.free()
    return out
