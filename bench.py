#!/usr/bin/env python3
"""bench.py -- CSR SpMV GB/s (% of HBM peak) + Jacobi-PCG iterations/s, 7-pt Poisson, fp64.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one y = A x with the full CSR operator already resident in HBM (N > 1: one
ghost exchange of x over RCCL + the local SpMV of every rank).  W untimed warm-up steps,
then exactly K steps between barrier + synchronize on both sides; MAX over ranks; rank 0
prints ONE JSON line.  value = algorithmic bytes of all ranks / that time, where the
algorithmic bytes of one SpMV are 12*nnz + 20*n + 4.

  N = 1   : 512^3 grid (BASELINE.json configs[2]): n = 134 217 728, nnz = 937 951 232
  N >= 2  : 1024 x 1024 x (128*N) grid split into z-slabs of 128 planes = 2^27 rows per GPU
            (N = 8 is the 1024^3 problem of configs[3]); weak scaling.

Beside it (not part of `value`): Jacobi-PCG iterations/s on the same operator (b = A*ones,
x0 = 0, fixed iteration count), the live HIP-event average of the SpMV kernel for the
`roofline` object, and -- rank 0, N = 1 only -- the CPU oracle (oracle/, single thread,
the reference's algorithm) timed on a bounded sample for the `cpu_baseline` object.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3-6.8 achievable)
W3_VARIANT = (1 << 20) + 128 + 64 + 2 + (64 << 8)  # csr_spmv_w3 (general banded CSR), see psp_csr.hip


def dram_model_bytes(kernel, info, n, nnz):
    """distinct bytes one launch has to move from/to DRAM: x and y once,
    plus the matrix stream of the kernel that ran"""
    if kernel == "csr_spmv_w4":   # values in padded offset-major blocks of 128 rows + 16-bit row masks
        rows = (n + 127) // 128 * 128
        return 8 * rows * info["nb"] + 2 * n + 16 * n
    if kernel == "csr_spmv_w3":   # val 8 + col16 2 per nonzero; per chunk of ~1016 nonzeros: block list + row offsets
        chunks = nnz / 1016.0
        return int(10 * nnz * (1024 / 1016.0) + chunks * (4 * info["nb"] + 2 * 256 + 16) + 16 * n)
    return 12 * nnz + 20 * n + 4


def spmv_bytes(n, nnz):
    return 12 * nnz + 20 * n + 4


def pcg_bytes(n, nnz):
    return 12 * nnz + 108 * n  # fused lower bound per iteration


class Events:
    """HIP events on the library's stream (the stream the kernels are launched on)."""

    def __init__(self, L, check):
        self.L, self.check = L, check
        self.e0, self.e1 = C.c_void_p(), C.c_void_p()
        check(L.psp_event_create(C.byref(self.e0)))
        check(L.psp_event_create(C.byref(self.e1)))

    def start(self):
        self.check(self.L.psp_event_record(self.e0))

    def stop_ms(self):
        self.check(self.L.psp_event_record(self.e1))
        ms = C.c_float()
        self.check(self.L.psp_event_elapsed_ms(self.e0, self.e1, C.byref(ms)))
        return float(ms.value)


def cpu_baseline(sample_n=256, spmv_reps=5, pcg_iters=10):
    """The oracle (C restatement of csr_mat.c:49-54 + pcg.c, gcc -O2, ONE thread) on a
    bounded sample: 7-pt Poisson sample_n^3 (same stencil, same bytes per row)."""
    from oracle import oracle as O
    t0 = time.time()
    A = O.poisson_csr(sample_n, sample_n, sample_n)
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
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    dinv = O.jacobi_dinv(A.diagonal())
    xs = np.zeros(n)
    t = time.perf_counter()
    O.pcg(A, b, xs, 0.0, pcg_iters, dinv)
    t_pcg = (time.perf_counter() - t) / (pcg_iters + 1)  # + the initial residual SpMV
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": spmv_bytes(n, nnz) / t_spmv / 1e9, "unit": "GB/s", "cores": 1, "kind": "port",
        "sample": "7-pt Poisson %d^3 (n=%d, nnz=%d): median of %d SpMV; %d Jacobi-PCG iterations"
                  % (sample_n, n, nnz, spmv_reps, pcg_iters),
        "pcg_iters_per_s": 1.0 / t_pcg, "host_cpu": model, "host_nproc": os.cpu_count(),
        "seconds": time.time() - t0,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--pcg-iters", type=int, default=40)
    ap.add_argument("--grid", default="", help="override the grid, e.g. 256,256,256 (testing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sss", action="store_true", help="skip the sss_mat leg (N = 1)")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--force-dist", action="store_true",
                    help="use the torch.distributed driver even at world size 1 (plumbing check)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch N > 1 through torch.distributed.run (one rank per GPU)")
        a.gpus = world

    use_dist = world > 1 or a.force_dist
    if use_dist or os.environ.get("PSP_IMPORT_TORCH"):
        # torch first: its bundled HIP runtime must be the one libpysparse_hip.so binds to --
        # two HIP runtimes in one process do not both see the GPU
        import torch
        import torch.distributed as dist
    from pysparse_amd import _capi, device as dev
    L, check = _capi.lib(), _capi.check

    if a.grid:
        nx, ny, nz = (int(t) for t in a.grid.split(","))
    elif world == 1:
        nx = ny = nz = 512
    else:
        nx = ny = 1024
        nz = 128 * world

    if use_dist:
        from pysparse_amd import distributed as D
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        be = D.HipBackend(local_rank)
        comm = D.Comm()
        A = D.DistCSR.poisson(nx, ny, nz, comm, be, dev.DeviceCSR.poisson_slab)
        A.A.set_variant(a.variant)
        n_loc, nnz_loc = A.A.shape[0], A.A.nnz
        x_ext = A.new_ext()
        A.owned(x_ext).copy_(be.from_numpy(np.random.default_rng(rank).standard_normal(n_loc)))
        y = be.zeros(n_loc)

        def step():
            A.matvec(x_ext, y)

        def sync():
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
    else:
        check(L.psp_set_device(local_rank))
        A = dev.DeviceCSR.poisson(nx, ny, nz)
        A.set_variant(a.variant)
        n_loc, nnz_loc = A.shape[0], A.nnz
        xb = dev.DeviceBuffer.from_host(np.random.default_rng(0).standard_normal(n_loc))
        yb = dev.DeviceBuffer(n_loc)

        def step():
            A.matvec_dev(xb.ptr, yb.ptr)

        def sync():
            check(L.psp_synchronize())

    ev = Events(L, check)
    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    ev.start()
    for _ in range(a.steps):
        step()
    ev_ms = ev.stop_ms()
    sync()
    wall = time.perf_counter() - t0

    # ---- beside it (N = 1): the general-CSR kernel on the same operator.  The default kernel for
    # a stencil operator (csr_spmv_w4) reads no column indices at all, so its rate in CSR-model
    # bytes can exceed the HBM line; csr_spmv_w3 is what an arbitrary banded csr_mat gets.
    general = None
    if not use_dist and a.variant < 0:
        kern0, info0 = A.kernel_info()
        if kern0 == "csr_spmv_w4":
            A.set_variant(W3_VARIANT)
            for _ in range(3):
                step()
            sync()
            ev.start()
            for _ in range(a.steps):
                step()
            g_ms = ev.stop_ms() / a.steps
            sync()
            general = {"kernel": A.kernel_info()[0], "avg_launch_ms": g_ms,
                       "achieved": spmv_bytes(n_loc, nnz_loc) / (g_ms * 1e-3) / 1e9}
            general["frac"] = general["achieved"] / HBM_PEAK_GBPS
            A.set_variant(-1)

    # ---- Jacobi-PCG iterations/s on the same operator (b = A*ones, x0 = 0, tol = 0 so that
    # exactly k iterations run; the setup -- ||b||, r = b - A x0 -- is inside the timed
    # region, i.e. the rate is slightly conservative)
    k = a.pcg_iters
    if use_dist:
        ones = A.new_ext()
        ones.fill_(1.0)
        b = be.zeros(n_loc)
        A.matvec(ones, b)
        dinv = be.zeros(n_loc)
        dinv.fill_(1.0 / (6.0 if nz > 0 else 4.0))  # constant diagonal of the Poisson operator
        for kk in (2, k):  # first call = warm-up
            xs = be.zeros(n_loc)
            sync()
            t = time.perf_counter()
            res = D.dist_pcg(A, b, xs, 0.0, kk, dinv)
            sync()
            pcg_t = time.perf_counter() - t
    else:
        K = dev.DeviceJacobi(A)
        aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
        bb = dev.DeviceBuffer(n_loc)
        ones_b = dev.DeviceBuffer.from_host(np.ones(n_loc))  # must outlive the asynchronous launch
        A.matvec_dev(ones_b.ptr, bb.ptr)
        sync()
        for kk in (2, k):
            xb.zero()
            info, it, rr = C.c_int(), C.c_int(), C.c_double()
            sync()
            t = time.perf_counter()
            check(L.psp_pcg_dev(aop._h, kop._h, n_loc, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it),
                                C.byref(rr), None))
            sync()
            pcg_t = time.perf_counter() - t
            res = (info.value, it.value, rr.value)
    pcg_s_per_iter = pcg_t / k

    # ---- beside it (N = 1): the same operator as an sss_mat (examples/poisson_test.py solves with
    # S = L.to_sss()): y = S x from the strict lower triangle only, and Jacobi-PCG on it
    sss = None
    if not use_dist and not a.no_sss:
        S = dev.DeviceSSS.poisson(nx, ny, nz)
        for _ in range(3):
            S.matvec_dev(xb.ptr, yb.ptr)
        sync()
        ev.start()
        for _ in range(a.steps):
            S.matvec_dev(xb.ptr, yb.ptr)
        s_ms = ev.stop_ms() / a.steps
        sync()
        nnz_lower = S.nnz - n_loc
        KS = dev.DeviceJacobi(S)
        sop, ksop = dev._Op(S, "matvec"), dev._Op(KS, "precon")
        for kk in (2, k):
            xb.zero()
            info, it, rr = C.c_int(), C.c_int(), C.c_double()
            sync()
            t = time.perf_counter()
            check(L.psp_pcg_dev(sop._h, ksop._h, n_loc, xb.ptr, bb.ptr, 0.0, kk, C.byref(info), C.byref(it),
                                C.byref(rr), None))
            sync()
            s_pcg_t = time.perf_counter() - t
        sss = {"kernel": S.kernel_info()[0], "spmv_ms": s_ms,
               # B_sss = 12 nnz_lower + 28 n + 4
               "spmv_GBps_sss_model": (12 * nnz_lower + 28 * n_loc + 4) / (s_ms * 1e-3) / 1e9,
               "pcg_iters_per_s": k / s_pcg_t, "pcg_check": {"info": info.value, "iter": it.value, "relres": rr.value}}
        sss["frac_sss_model"] = sss["spmv_GBps_sss_model"] / HBM_PEAK_GBPS
        del S, KS, sop, ksop

    # ---- MAX over ranks
    if use_dist:
        t = torch.tensor([wall, ev_ms, pcg_s_per_iter], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, ev_ms, pcg_s_per_iter = t.tolist()
        tot = torch.tensor([float(n_loc), float(nnz_loc)], dtype=torch.float64, device="cuda")
        dist.all_reduce(tot)
        n_tot, nnz_tot = (int(v) for v in tot.tolist())
    else:
        n_tot, nnz_tot = n_loc, nnz_loc

    if rank == 0:
        ms_step = wall * 1e3 / a.steps
        value = spmv_bytes(n_tot, nnz_tot) / (wall / a.steps) / 1e9
        kern_ms = ev_ms / a.steps
        achieved = spmv_bytes(n_loc, nnz_loc) / (kern_ms * 1e-3) / 1e9  # one GPU, one launch
        kernel, kinfo = (A.A if use_dist else A).kernel_info()
        # HBM traffic of one launch from the committed rocprofv3 --pmc passes of the same kernel
        # on the same workload (tools/make_profiles.sh; counters cannot be read in-process)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r1_spmv_pmc.json")
        if os.path.exists(pmc) and world == 1 and not a.grid:
            try:
                rec = json.load(open(pmc))
                if rec.get("kernel") == kernel:
                    traffic = rec.get("hbm_bytes_per_launch")
            except (OSError, ValueError):
                traffic = None
        out = {
            "metric": "CSR SpMV GB/s (7-pt Poisson, % of 8 TB/s HBM peak) + PCG iters/s",
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "3D Poisson 7-pt %dx%dx%d fp64 CSR (int32 indices), y = A x%s" % (
                    nx, ny, nz, "" if world == 1 else "; z-slab row partition, ghost exchange over RCCL"),
                "n": n_tot, "nnz": nnz_tot, "rows_per_gpu": n_loc,
                "parallelism": "1 GPU" if world == 1 else "row-range x%d" % world,
            },
            "pct_hbm_peak": 100.0 * value / (HBM_PEAK_GBPS * world),
            "pcg_iters_per_s": 1.0 / pcg_s_per_iter,
            "pcg_effective_GBps": pcg_bytes(n_tot, nnz_tot) / pcg_s_per_iter / 1e9,
            "pcg_check": {"info": res[0], "iter": res[1], "relres": res[2], "iters_timed": k},
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "algorithmic_bytes_per_launch": spmv_bytes(n_loc, nnz_loc), "avg_launch_ms": kern_ms,
                # what the kernel that ran has to move (its own index format), and that rate
                "dram_model_bytes_per_launch": dram_model_bytes(kernel, kinfo, n_loc, nnz_loc),
                "dram_model_GBps": dram_model_bytes(kernel, kinfo, n_loc, nnz_loc) / (kern_ms * 1e-3) / 1e9,
                "note": "achieved = CSR-model bytes (12 nnz + 20 n + 4) / avg launch time; "
                        "csr_spmv_w4/w3 move fewer bytes than that model (no / 16-bit column indices)",
            },
        }
        if general is not None:
            out["roofline_general_csr"] = general
        if sss is not None:
            out["sss_mat"] = sss
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
