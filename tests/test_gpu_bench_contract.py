"""GPU: bench.py prints ONE JSON line with the driver's contract (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config) plus the
`roofline` and `cpu_baseline` objects, on a small grid so that it runs in seconds.  No fraction of the
HBM peak may exceed 1 by construction: every rate is bytes the kernel that ran has to move / time."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(stdout):
    """ONE JSON line of at most bench_line.LINE_LIMIT characters on stdout; the full record it stands for is in the side
    file it names (round 6).  Returns the full record, with the line itself under "_line"."""
    sys.path.insert(0, ROOT)
    import bench_line
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    assert len(lines[0]) <= bench_line.LINE_LIMIT, len(lines[0])
    line, full = bench_line.read(stdout)
    for k in line:  # whatever the line says, the side file says too (the launcher object: in full there, in short here)
        if k not in ("side_file", "side_keys", "launcher", "config", "roofline", "cpu_baseline", "parity_check", "phases",
                     "parity_vs_n1", "preflight", "transport", "provenance", "published_table", "config5"):
            assert full.get(k) == line[k], k
    d = dict(full)
    d["_line"] = line
    return d


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True,
                         text=True, check=True, cwd=ROOT).stdout
    return parse(out)


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


@pytest.mark.gpu
def test_bench_json_contract_small_grid():
    d = run_bench("--grid", "96,96,96", "--steps", "5", "--warmup", "2", "--pcg-iters", "8", "--no-cpu-baseline",
                  "--no-clocks")
    for k in CONTRACT:
        assert k in d, k
    assert d["unit"] == "GB/s" and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] in ("strong", "weak") and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    n, nnz = d["config"]["n"], d["config"]["nnz"]
    assert n == 96 ** 3 and nnz == 7 * n - 6 * 96 * 96
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["kernel"] == "csr_spmv_w4"
    # what csr_spmv_w4 has to move: 7 offset slots of 8 B per (padded) row + 2 B mask + x + y
    assert r["algorithmic_bytes_per_launch"] == 8 * 7 * ((n + 127) // 128 * 128) + 18 * n
    assert r["csr_model_bytes_per_launch"] == 12 * nnz + 20 * n + 4
    assert r["algorithmic_bytes_per_launch"] < r["csr_model_bytes_per_launch"]  # no column indices are read
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    # nothing is priced above the peak
    assert r["frac"] <= 1.0 and d["pct_hbm_peak"] <= 100.0 and d["pcg_pct_hbm_peak"] <= 100.0
    ks = {k["kernel"]: k for k in d["kernels_same_operator"]}
    assert set(ks) == {"csr_spmv_w3", "csr_spmv_w6", "csr_spmv_w2"}
    for lit in ("csr_spmv_w6", "csr_spmv_w2"):  # both stream the CSR arrays as stored
        assert ks[lit]["bytes_per_launch"] == 12 * nnz + 20 * n + 4
    # the second reading of "CSR SpMV % of peak": the better of the two literal-CSR kernels, in SURVEY 8d's bytes
    assert r["streaming_kernel"] in ("csr_spmv_w6", "csr_spmv_w2")
    assert r["csr_model_frac_of_streaming_kernel"] == max(ks["csr_spmv_w6"]["csr_model_frac"], ks["csr_spmv_w2"]["csr_model_frac"])
    assert all(k["frac"] <= 1.0 for k in ks.values())
    assert d["sss_mat"]["kernel"] == "sss_spmv_w4" and d["sss_mat"]["frac"] <= 1.0
    # the reference's other Krylov solvers on the same operator, priced in the bytes their kernels move
    sv = d["solvers"]
    op_bytes = r["algorithmic_bytes_per_launch"] - 16 * n
    for name, prods, streams in (("cgs", 2, 21), ("bicgstab", 2, 24), ("qmrs", 1, 19)):
        assert sv[name]["bytes_per_iter"] == prods * op_bytes + streams * 8 * n and sv[name]["ms_per_iter"] > 0
        assert 0 < sv[name]["frac"] <= 1.0
    assert sv["gmres20"]["last"][1] == 60 and 0 < sv["gmres20"]["frac"] <= 1.0
    assert sv["cgs"]["last"][1] == 25 and sv["qmrs"]["last"][1] == 25
    assert d["pcg_check"]["info"] == -1 and d["pcg_check"]["iter"] == 9  # tol = 0: exactly 8 iterations
    assert d["pcg_iters_per_s"] > 0 and d["value"] > 0
    # the placement sweep rides along and never feeds `value`: the first allocation's launch time is roofline's
    ps = d["placement_sweep"]
    assert len(ps["y_realloc_avg_launch_ms"]) == 6 and ps["first_allocation_ms"] == r["avg_launch_ms"]
    assert 0 < ps["best_ms"] <= ps["worst_ms"] and ps["best_frac_of_peak"] <= 1.0
    # roofline.traffic: measured by rocprofv3 --pmc child runs of the same operator in this job -- where the box lets
    # an ordinary user read the counters; otherwise null (no committed pass exists for this grid), never a guess
    if r["traffic_source"] and r["traffic_source"].startswith("measured in this job"):
        assert 0.5 * r["algorithmic_bytes_per_launch"] <= r["traffic"] <= 2.0 * r["algorithmic_bytes_per_launch"]
        assert r["traffic_counters"]["fetch_correction"] == 2.0
    else:
        assert r["traffic"] is None
    c = d["device_ceiling_same_run"]
    assert c["read_only_dot"]["bytes"] == 16 * n and c["read2_write1"]["bytes"] == 24 * n
    assert c["read_only_dot"]["GBps"] > 0 and c["read2_write1"]["GBps"] > 0


@pytest.mark.gpu
def test_bench_strong_path_world_size_1():
    """--gpus 1 --scaling strong: the multi-GPU driver (index-free slab operator, RCCL process group of one rank)"""
    d = run_bench("--gpus", "1", "--scaling", "strong", "--grid", "64,64,64", "--steps", "5", "--warmup", "2",
                  "--pcg-iters", "8", "--no-cpu-baseline", "--no-clocks")
    for k in CONTRACT:
        assert k in d, k
    assert d["scaling"] == "strong" and d["rccl_ranks"] == 1 and d["backend"] == "nccl"
    assert d["roofline"]["kernel"] == "csr_spmv_w4" and d["roofline"]["frac"] <= 1.0
    assert d["pcg_check"]["info"] == -1 and d["pcg_check"]["iter"] == 9


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_bench_n_ranks_rehearsal_on_one_gpu(world):
    """`python bench.py --gpus N` end to end with the HIP kernels: self-launched ranks, index-free slab operators,
    halo exchange of device tensors, device-resident scalars with in-stream all-reduces -- everything of the N > 1
    path except the transport, which is gloo here (it moves device tensors on this image; RCCL needs one GPU per
    rank and the box has one).  World 3: the middle rank has halos on both sides."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--backend", "gloo",
                          "--share-gpu", "--grid", "64,48,36", "--steps", "4", "--warmup", "2", "--pcg-iters", "40",
                          "--no-cpu-baseline", "--no-clocks"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    d = parse(out.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == world and d["rccl_ranks"] == world and d["backend"] == "gloo" and "dry_run" in d
    assert d["scaling"] == "strong" and d["config"]["n"] == 64 * 48 * 36
    assert d["launcher"]["stage"] == "torch_rccl_ranks" and d["launcher"]["fallback_from"] == []
    # the in-job parity check: 20 iterations of the N-rank solve against rank 0's one-GPU solve of the whole problem
    assert d["parity_vs_n1"]["ok"] and d["parity_vs_n1"]["max_rel_diff"] <= 1e-9, d["parity_vs_n1"]
    assert d["strong_n1"]["grid"] == [64, 48, 36] and d["vs_n1"] > 0
    assert d["phases"]["iteration_ms"] > 0 and d["phases"]["halo_ms"] > 0 and 0.0 <= d["phases"]["overlap_frac"] <= 1.0
    assert d["preflight"]["peer_access"] == [[1]]
    assert d["roofline"]["kernel"] == "csr_spmv_w4"
    # tol = 0: exactly 40 iterations; the same count and residual as the one-GPU solver on the whole problem
    assert d["pcg_check"]["info"] == -1 and d["pcg_check"]["iter"] == 41
    sys.path.insert(0, ROOT)
    import numpy as np
    from pysparse_amd import device as dev
    A = dev.DeviceCSR.poisson(64, 48, 36)
    n = A.shape[0]
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    x = np.zeros(n)
    ref = dev.pcg(A, b, x, 0.0, 40, dev.DeviceJacobi(A))
    assert ref[:2] == (-1, 41) and abs(ref[2] - d["pcg_check"]["relres"]) <= 1e-9 * ref[2]


def test_bench_cpu_baseline_objects_small_sample():
    sys.path.insert(0, ROOT)
    import bench
    base, ref, parity = bench.cpu_baseline(c2_grid=(40, 40, 0), c3_grid=(24, 24, 24), c3_small=(16, 16, 16))
    assert parity is None  # no device handed over: the GPU-against-oracle comparison is test_gpu_reference_sizes.py / the bench line
    assert base["kind"] == "port" and base["cores"] == 1 and base["unit"] == "GB/s" and base["value"] > 0
    assert base["pcg_iters_per_s"] > 0 and "C2_poisson2d_40" in base
    if ref is not None:  # oracle/_ref is built where /root/reference exists
        assert ref["kind"] == "reference" and ref["cores"] == 1 and ref["value"] > 0
        assert ref["iterates_match_port"] and ref["iterates_max_rel_diff"] <= ref["iterates_tolerance"]


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus 2` (no torch.distributed environment) must start 2 fresh ranks by itself and
    print ONE JSON line: launcher + row-range driver over gloo with the oracle-backed test backend."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "12,10,8",
                          "--steps", "3", "--warmup", "1", "--pcg-iters", "5", "--no-cpu-baseline",
                          "--test-backend", "tests.dist_oracle_backend:bench_factory"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = parse(out.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "strong" and "dry_run" in d
    assert d["config"]["n"] == 12 * 10 * 8 and d["config"]["rows_per_gpu"] == 12 * 10 * 4
    assert d["pcg_check"]["info"] == -1 and d["pcg_check"]["iter"] == 6
    # the line names the stage of the ladder that produced it, the transport, and what the run was checked against
    assert d["launcher"]["stage"] == "torch_rccl_ranks" == d["stage"] and d["launcher"]["fallback_from"] == []
    assert d["launcher"]["ladder"] == ["torch_rccl_ranks", "single_process_rccl", "single_process_fold"]
    assert "halo" in d["transport"] and "reductions" in d["transport"]
    par = d["parity_vs_n1"]  # PARITY_ITERS iterations of the 2-rank solve against the unpartitioned solve
    assert par["ok"] and par["iters"] == 20 and par["max_rel_diff"] <= 1e-9 and par["same_info_iter"]
    ph = d["phases"]
    for k in ("px_update_ms", "spmv_interior_ms", "halo_exposed_ms", "spmv_boundary_ms", "allreduce_1_ms", "scalar_1_ms",
              "r_update_ms", "allreduce_2_ms", "scalar_2_ms", "halo_ms", "iteration_ms"):
        assert ph[k] >= 0.0, k
    assert len(ph["allreduce_us"]) == 2 and 0.0 <= ph["overlap_frac"] <= 1.0
    assert d["preflight"]["world"] == 2
    assert d["provenance"]["match"] is True


def test_bench_eight_ranks_dry_run_is_the_shape_of_the_scaling_run():
    """the driver's scaling run ends at N = 8: the same launcher + row-range driver over gloo with the oracle-backed test
    backend at eight ranks (a 16^3 grid: two planes per rank, every interior rank with halos on both sides) -- the line stays
    under the limit, names its stage, passes its in-job parity check and carries the prediction it will be judged against"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--grid", "16,16,16",
                          "--steps", "3", "--warmup", "1", "--pcg-iters", "5", "--no-cpu-baseline",
                          "--test-backend", "tests.dist_oracle_backend:bench_factory"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = parse(out.stdout)
    line = d["_line"]
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "strong" and "dry_run" in d
    assert d["config"]["n"] == 4096 and d["config"]["rows_per_gpu"] == 512
    assert d["launcher"]["stage"] == "torch_rccl_ranks" and d["launcher"]["fallback_from"] == []
    assert d["parity_vs_n1"]["ok"] and d["parity_vs_n1"]["max_rel_diff"] <= 1e-9
    assert d["pcg_check"]["info"] == -1 and d["pcg_check"]["iter"] == 6
    p = line["predicted"]
    assert p["allreduce_us"] == [30.0, 30.0] and p["halo_exposed_ms"] == 0.0 and "missed_budget" in p
    assert line["roofline"]["pcg_iters_per_s"] == d["pcg_iters_per_s"] > 0 and line["phases"]["iteration_ms"] > 0
    assert d["preflight"]["world"] == 8


def _run_ladder(*extra, timeout=240):
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "12,10,8",
                          "--steps", "3", "--warmup", "1", "--pcg-iters", "5", "--no-cpu-baseline",
                          "--test-backend", "tests.dist_oracle_backend:bench_factory"] + list(extra),
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    assert len([l for l in out.stdout.strip().splitlines() if l.startswith("{")]) == 1, out.stdout + out.stderr[-2000:]
    return out.returncode, parse(out.stdout), time.time() - t0


def test_bench_ladder_a_rank_that_exits_gives_an_error_line_and_a_nonzero_exit_code():
    """first contact with N > 1 must not be all-or-nothing: a rank that dies after the process group has formed ends
    the stage, and with no stage left the job prints ONE JSON error line (value null) and leaves with rc != 0"""
    rc, d, wall = _run_ladder("--inject", "exit:1", "--ladder", "torch_rccl_ranks")
    assert rc != 0 and d["value"] is None and "error" in d and d["n_gpus"] == 2
    f = d["launcher"]["fallback_from"]
    assert len(f) == 1 and f[0]["stage"] == "torch_rccl_ranks" and f[0]["rc"] != 0
    assert any("injected failure: rank 1 exits" in l for l in f[0]["stderr_tail"])
    assert wall < 120


def test_bench_ladder_a_rank_that_hangs_is_ended_by_the_stage_timeout():
    """a rank that never arrives (a hang in communicator set-up looks like this): the others block in the first
    collective, the stage's time-out ends the whole process group, the line still comes, inside the limit"""
    rc, d, wall = _run_ladder("--inject", "hang:1", "--ladder", "torch_rccl_ranks", "--stage-timeout", "25")
    assert rc != 0 and d["value"] is None and "error" in d
    f = d["launcher"]["fallback_from"]
    assert len(f) == 1 and "timed out" in f[0]["reason"]
    assert 25 <= wall < 90


def test_bench_ladder_falls_through_every_stage_and_respects_the_deadline():
    """all three stages: the torch ranks fail by injection; the two single-process stages need GPUs this container does
    not have and say so; the error line lists what each stage died of"""
    from pysparse_amd import device
    if device.device_count() > 0:
        pytest.skip("a GPU is present: the single-process stages would run")
    rc, d, wall = _run_ladder("--inject", "exit:0", "--deadline", "200")
    assert rc != 0 and d["value"] is None
    f = d["launcher"]["fallback_from"]
    assert [x["stage"] for x in f] == ["torch_rccl_ranks", "single_process_rccl", "single_process_fold"]
    assert any("no HIP device" in l for l in f[1]["stderr_tail"])
    assert wall < 200
    # a deadline too short for anything: every stage is skipped, the line still comes
    rc, d, wall = _run_ladder("--deadline", "10")
    assert rc != 0 and all(x["reason"].startswith("skipped") for x in d["launcher"]["fallback_from"])


@pytest.mark.gpu
def test_bench_ladder_falls_back_to_the_single_process_stage_on_the_gpu():
    """the ladder end to end on the one-GPU box: the torch ranks (gloo, sharing cuda:0) die by injection, the next stage --
    ONE process, the device list through the C ABI -- produces the line, which says so; its 3-rank solve agrees with the
    one-GPU solve of the same system (parity_vs_n1)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--backend", "gloo",
                          "--share-gpu", "--grid", "64,48,36", "--steps", "4", "--warmup", "2", "--pcg-iters", "24",
                          "--no-cpu-baseline", "--no-clocks", "--inject", "exit:2"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = parse(out.stdout)
    assert d["launcher"]["stage"] == "single_process_rccl" == d["stage"]
    f = d["launcher"]["fallback_from"]
    assert len(f) == 1 and f[0]["stage"] == "torch_rccl_ranks"
    assert d["ranks"] == 3 and d["reductions"].startswith("fold kernel")  # three ranks on ONE device: no RCCL
    assert d["parity_vs_n1"]["ok"] and d["parity_vs_n1"]["max_rel_diff"] <= 1e-9
    assert d["strong_n1"]["grid"] == [64, 48, 36] and d["vs_n1"] > 0
    for k in ("halo_ms", "spmv_local_ms", "allreduce_ms", "spmv_with_halo_ms", "allreduce_us"):
        assert d["phases"][k] >= 0.0
    assert d["preflight"]["peer_access"] == [[1]] and d["provenance"]["match"] is True


@pytest.mark.gpu
def test_bench_single_process_device_list_rehearsal():
    """`bench.py --gpus N --single-process`: the N-GPU job as ONE process through psp_csr_poisson_multi (the C ABI's
    device-list variant, pysparse_amd/csrc/psp_multi.hip).  On this one-GPU box --share-gpu lists device 0 three times:
    every piece of the path but peer copies / RCCL between different devices; the line says so (dry_run)."""
    d = run_bench("--gpus", "3", "--single-process", "--share-gpu", "--grid", "64,64,66", "--steps", "5", "--warmup", "2",
                  "--pcg-iters", "16")
    for k in CONTRACT[:-1]:
        assert k in d, k
    assert d["parity_vs_n1"]["ok"] and d["phases"]["spmv_with_halo_ms"] > 0
    assert d["n_gpus"] == 3 and d["ranks"] == 3 and d["distinct_devices"] == 1 and d["scaling"] == "strong"
    assert d["config"]["n"] == 64 * 64 * 66 and d["config"]["devices"] == [0, 0, 0]
    assert "dry_run" in d and d["reductions"].startswith("fold kernel")
    assert d["pcg_check"]["info"] == -1 and d["pcg_check"]["iter"] == 4 + 16 + 1  # tol = 0: exactly k2 iterations
    assert d["pcg_iters_per_s"] > 0 and d["value"] > 0 and d["pct_hbm_peak"] <= 100.0
    one = run_bench("--gpus", "1", "--single-process", "--grid", "64,64,66", "--steps", "5", "--warmup", "2",
                    "--pcg-iters", "16")
    assert one["ranks"] == 1 and one["reductions"] == "none" and "dry_run" not in one
    # same right-hand side, same iteration count: the recurred residual agrees to rounding between 1 and 3 ranks
    assert abs(one["pcg_check"]["relres"] - d["pcg_check"]["relres"]) <= 1e-9 * one["pcg_check"]["relres"]


@pytest.mark.gpu
def test_bench_mtx_leg_on_a_matrix_market_file(tmp_path):
    """`bench.py --mtx FILE` -- BASELINE.json configs[4] for a matrix the user supplies (the day Emilia_923.mtx is at hand):
    ingest, kernel chosen, SpMV in SSS- and CSR-model bytes, Jacobi-MINRES us/iteration, parity against the oracle.  Here
    on a small symmetric FEM-like file written on the spot."""
    sys.path.insert(0, ROOT)
    from pysparse_amd.tools import standins
    n, ind, col, val, diag = standins.fem_sss_arrays(14, 13, 12, 16)
    path = os.path.join(str(tmp_path), "fem_small.mtx")
    import numpy as np
    r = np.repeat(np.arange(n), np.diff(ind))
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate real symmetric\n")
        f.write("%d %d %d\n" % (n, n, val.size + n))
        for i in range(n):
            f.write("%d %d %.17g\n" % (i + 1, i + 1, diag[i]))
        for i, j, v in zip(r, col, val):
            f.write("%d %d %.17g\n" % (i + 1, j + 1, v))
    d = run_bench("--mtx", path, "--steps", "10")
    for k in CONTRACT:
        assert k in d, k
    c = d["config5"]
    assert c["n"] == n and c["nnz_lower"] == val.size and c["nnz_full"] == 2 * val.size + n
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 12 * val.size + 28 * n + 4 == c["sss_model_bytes"]
    assert c["csr_model_bytes"] == 12 * (2 * val.size + n) + 20 * n + 4
    assert c["parity"]["ok"] and c["parity"]["spmv_bit_exact_vs_oracle"] and c["parity"]["x_max_rel_diff"] <= 1e-12
    assert c["minres"]["info"] == 0 and c["minres"]["us_per_iteration"] > 0 and c["kernel"]
    assert 0 < d["roofline"]["frac"] <= 1.0 and c["csr_model_frac_of_peak"] <= 1.0
    assert "error" not in d and d["data"] == "user file"
    # round 6: time to solution (the handle's cost rule) beside the steady state (products announced)
    t = c["time_to_solution"]
    assert t["end_to_end_ms"] >= t["upload_ms"] + t["first_solve_ms"] > 0 and c["cold"]["kernel"] and c["setup_ms"] >= 0.0
    assert c["kernel_info"]["setup_ms"] == c["setup_ms"] and c["parity"]["x_max_rel_diff_renumbered"] <= 1e-12
    l5 = d["_line"]["config5"]
    assert l5["parity_ok"] is True and l5["time_to_solution"]["end_to_end_ms"] == t["end_to_end_ms"]


def _run_external(*extra, timeout=240):
    """bench.py as ONE RANK OF N under somebody else's torch.distributed.run -- how the driver's scaling run starts it"""
    import socket
    import time
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--grid", "12,10,8", "--steps", "3", "--warmup", "1", "--pcg-iters", "5",
                          "--no-cpu-baseline", "--test-backend", "tests.dist_oracle_backend:bench_factory"] + list(extra),
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    return out.returncode, lines, time.time() - t0, out.stderr


def test_bench_as_a_rank_of_an_external_launcher_prints_its_line():
    rc, lines, wall, err = _run_external()
    assert rc == 0 and len(lines) == 1, err[-2000:]
    d = parse(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["parity_vs_n1"]["ok"] and d["pcg_check"]["iter"] == 6


def test_bench_rank_guard_a_hanging_rank_still_ends_in_one_json_line():
    """the driver starts the ranks itself (`python -m torch.distributed.run ... bench.py --gpus N`), so the ladder of
    `python bench.py --gpus N` is not in play: every rank guards itself.  A rank that hangs lets its own watchdog leave
    quietly and note why; rank 0 sees the note, runs the single-process stages as fresh children (they need GPUs this
    container lacks and say so) and prints ONE error line; the job ends with rc != 0 well inside the limit"""
    from pysparse_amd import device
    if device.device_count() > 0:
        pytest.skip("a GPU is present: rank 0's fall-back stages would run")
    rc, lines, wall, err = _run_external("--inject", "hang:1", "--rank-deadline", "12")
    assert rc != 0 and len(lines) == 1, err[-3000:]
    d = parse(lines[0])
    assert d["value"] is None and "error" in d and d["n_gpus"] == 2
    f = d["launcher"]["fallback_from"]
    assert [x["stage"] for x in f] == ["torch_rccl_ranks", "single_process_rccl", "single_process_fold"]
    assert "hangs" in f[0]["reason"]  # rank 0's own deadline, or rank 1's note, whichever it sees first
    assert wall < 120


def test_bench_rank_guard_answers_the_launchers_sigterm_with_an_error_line():
    """a rank that crashes hard makes the launcher end the others: rank 0 answers the SIGTERM with the error line"""
    rc, lines, wall, err = _run_external("--inject", "exit:1")
    assert rc != 0 and len(lines) == 1, err[-3000:]
    d = parse(lines[0])
    assert d["value"] is None and "SIGTERM" in d["launcher"]["fallback_from"][0]["reason"]


@pytest.mark.gpu
def test_bench_rank_guard_falls_back_to_the_single_process_stage_on_the_gpu():
    """ranks started by an external torch.distributed.run (the driver's way), one of them hangs: its watchdog leaves
    quietly, rank 0 runs the single-process stage as a fresh child on the GPU and prints THAT line, rc 0"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--backend", "gloo", "--share-gpu", "--grid", "64,48,36", "--steps", "4",
                          "--warmup", "2", "--pcg-iters", "24", "--no-cpu-baseline", "--no-clocks", "--inject", "hang:1",
                          "--rank-deadline", "12"],
                         capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = parse(out.stdout)
    assert d["launcher"]["stage"] == "single_process_rccl" and "external launcher" in d["launcher"]["started_by"]
    assert d["launcher"]["fallback_from"][0]["stage"] == "torch_rccl_ranks" and "hangs" in d["launcher"]["fallback_from"][0]["reason"]
    assert d["ranks"] == 2 and d["parity_vs_n1"]["ok"] and d["value"] > 0


@pytest.mark.gpu
def test_bench_ladder_survives_a_real_rccl_failure():
    """not an injected failure: two torch ranks over RCCL on ONE GPU -- RCCL refuses (duplicate device) and the ranks die with a
    DistBackendError within seconds; the ladder's next stage (one process, device list, the GPU listed twice) produces the line"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--grid", "64,48,36",
                          "--steps", "4", "--warmup", "2", "--pcg-iters", "24", "--no-cpu-baseline", "--no-clocks",
                          "--stage-timeout", "150"], capture_output=True, text=True, cwd=ROOT, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = parse(out.stdout)
    assert d["launcher"]["stage"] == "single_process_rccl"
    f = d["launcher"]["fallback_from"]
    assert len(f) == 1 and f[0]["stage"] == "torch_rccl_ranks" and f[0]["rc"] != 0
    assert any("NCCL" in l or "DistBackendError" in l for l in f[0]["stderr_tail"]), f[0]["stderr_tail"]
    assert d["parity_vs_n1"]["ok"] and d["ranks"] == 2
