"""bench.py's contract-critical part on its own (round 5, VERDICT r4 "Next" #8): the timed region and the numbers of the
JSON line that come out of it.  No GPU: the step, the synchronisation and the events are fakes that record what was done.

  * exactly W untimed and K timed steps; one synchronisation before the clock starts and one before it stops; the two
    events are recorded around the K timed launches, inside the bracket;
  * value = bytes of the whole job per step / wall time per step; roofline.achieved = one GPU's bytes per launch / the
    average launch between the events -- recomputed here from the fakes' numbers;
  * nothing in the region, in the functions it calls, or in the launcher / legs modules can reach the oracle: the oracle
    is imported in bench.py only, inside cpu_baseline / gpu_parity_case / _ref_solve / mtx_leg (the CPU-baseline and
    parity legs), never at module level."""
import ast
import os
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeEvents:
    def __init__(self, log, ms):
        self.log, self._ms = log, ms

    def record(self, i):
        self.log.append(("record", i))

    def ms(self, i, j):
        assert (i, j) == (0, 1)
        return self._ms


@pytest.mark.parametrize("warmup,steps", [(0, 1), (3, 7), (10, 100)])
def test_timed_region_times_exactly_k_steps_between_two_synchronisations(warmup, steps):
    import bench
    log = []

    def step():
        log.append(("step",))
        time.sleep(0.0002)

    def sync():
        log.append(("sync",))
    wall, ev_ms = bench.timed_region(step, sync, FakeEvents(log, 12.5), warmup, steps)
    want = [("step",)] * warmup + [("sync",), ("record", 0)] + [("step",)] * steps + [("record", 1), ("sync",)]
    assert log == want
    assert ev_ms == 12.5
    assert wall >= steps * 0.0002 and wall < steps * 0.0002 + 0.5  # the K timed steps, not the warm-up
    # no device (gloo dry runs): the event time is the wall time
    log.clear()
    wall2, ev2 = bench.timed_region(step, sync, None, warmup, steps)
    assert ev2 == wall2 * 1e3 and ("record", 0) not in log


def test_headline_numbers_are_bytes_over_the_measured_times():
    import bench
    kb_job, kb_launch, wall, ev_ms, steps = 8 * 9932111872, 9932111872, 0.0335, 33.1, 20
    h = bench.headline(kb_job, kb_launch, wall, ev_ms, steps)
    assert h["value"] == kb_job / (wall / steps) / 1e9
    assert h["ms_per_step"] == wall * 1e3 / steps
    assert h["avg_launch_ms"] == ev_ms / steps
    assert h["achieved"] == kb_launch / (ev_ms / steps * 1e-3) / 1e9
    assert h["frac"] == h["achieved"] / 8000.0
    # one GPU: the job's bytes are the launch's bytes, and the two rates differ only by wall vs event time
    h1 = bench.headline(kb_launch, kb_launch, wall, ev_ms, steps)
    assert abs(h1["value"] / h1["achieved"] - (ev_ms * 1e-3) / wall) < 1e-12


def _imports_of(node):
    out = []
    for n in ast.walk(node):
        if isinstance(n, ast.Import):
            out += [a.name for a in n.names]
        elif isinstance(n, ast.ImportFrom):
            out.append(n.module or "")
    return out


def test_the_oracle_is_reachable_from_the_baseline_and_parity_legs_only():
    src = {f: open(os.path.join(ROOT, f)).read() for f in ("bench.py", "bench_common.py", "bench_launch.py", "bench_legs.py",
                                                            "bench_line.py")}
    # the launcher, the legs, the line and the shared module never mention it
    for f in ("bench_common.py", "bench_launch.py", "bench_legs.py", "bench_line.py"):
        assert not any("oracle" in m for m in _imports_of(ast.parse(src[f]))), f
    tree = ast.parse(src["bench.py"])
    allowed = {"cpu_baseline", "mtx_leg"}  # gpu_parity_case / _cpu_case / _ref_solve are handed the module by these
    for node in tree.body:
        imps = [m for m in _imports_of(node) if "oracle" in m]
        if isinstance(node, ast.FunctionDef):
            assert not imps or node.name in allowed, (node.name, imps)
        else:
            assert not imps, "module-level oracle import in bench.py"
    fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    # the region itself: no imports at all, and it calls only what it is handed (+ the clock)
    region = fns["timed_region"]
    assert _imports_of(region) == []
    called = {ast.unparse(c.func) for c in ast.walk(region) if isinstance(c, ast.Call)}
    assert called <= {"step", "sync", "ev.record", "ev.ms", "time.perf_counter", "range"}, called
    assert _imports_of(fns["headline"]) == []
    # the body of a rank reaches the oracle only through cpu_baseline(...) / mtx_leg(...), both after the timed region
    body = fns["run_body"]
    lines = src["bench.py"].splitlines()
    region_line = next(c.lineno for c in ast.walk(body) if isinstance(c, ast.Call) and ast.unparse(c.func) == "timed_region")
    for c in ast.walk(body):
        if isinstance(c, ast.Call) and ast.unparse(c.func) in ("cpu_baseline", "mtx_leg"):
            assert c.lineno > region_line, lines[c.lineno - 1]


# ---------------------------------------------------------------------------------------------- the printed line (round 6)

def _full_record(world=1):
    """a full record of the size and shape run_body assembles (numbers from BENCH_r05), bulky legs included"""
    n, nnz = 134217728, 937951232
    filler = {"note": "x" * 900, "table": list(range(300))}
    full = {
        "metric": "CSR SpMV GB/s (7-pt Poisson, % of 8 TB/s HBM peak) + PCG iters/s", "value": 6027.5, "unit": "GB/s",
        "n_gpus": world, "steps": 20, "warmup": 5, "ms_per_step": 1.6478, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3D Poisson 7-pt 512x512x512 fp64 csr_mat (int32 indices), y = A x", "n": n, "nnz": nnz,
                   "rows_per_gpu": n // world, "parallelism": "1 GPU", "scaling_mode": "single"},
        "pct_hbm_peak": 75.3, "effective_csr_model_GBps": 8459.0, "pcg_iters_per_s": 327.1,
        "pcg_check": {"info": -1, "iter": 101, "relres": 1e-3, "loop": "pcg_lazy_pf"},
        "roofline": {"bound": "hbm", "kernel": "csr_spmv_w4", "achieved": 6032.2, "peak": 8000.0, "unit": "GB/s",
                     "frac": 0.754, "traffic": 11020000000, "traffic_source": "measured in this job: " + "y" * 400,
                     "traffic_counters": filler, "algorithmic_bytes_per_launch": 9932111872, "avg_launch_ms": 1.6465,
                     "median_launch_ms": 1.646, "csr_model_bytes_per_launch": 13939769348, "note": "z" * 700,
                     "frac_8d_of_timed_kernel": 1.058, "frac_8d_note": "format-compressed kernel priced in CSR bytes",
                     "csr_literal": {"kernel": "csr_spmv_w6", "avg_launch_ms": 2.41, "frac_8d": 0.723},
                     "pcg_iters_per_s": 327.1, "pcg_loop": "pcg_lazy_pf", "pcg_launches_per_iter": 6,
                     "pcg_bytes_per_iter": 130 * n, "pcg_frac_own_bytes": 0.713, "strong_n1_iters_per_s": 40.3,
                     "stream_ceiling_GBps": 5986.0, "frac_of_stream_ceiling": 1.0, "placement_level": "slow"},
        "provenance": {"build_id": "0b07de8a42244a10", "source_hash": "0b07de8a42244a10", "match": True},
        "kernels_same_operator": [filler] * 3, "sss_mat": filler, "solvers": {k: filler for k in "abcd"},
        "single_kernel_loops": filler, "strong_n1": dict(filler, grid=[1024] * 3, pcg_iters_per_s=40.3),
        "placement_sweep": filler, "device_ceiling_same_run": filler, "process_mode": filler, "gpu_clocks_under_load": filler,
        "parity_check": {"ok": True, "what": "w" * 500, "cases": {
            "C3_poisson3d_512": {"pcg": {"vs_reference_module_kernel": {"x_max_rel_diff": 1.3e-13}}, "minres": {}, "ok": True}}},
        "cpu_baseline": {"value": 30.4, "unit": "GB/s", "cores": 1, "kind": "port", "sample": "s" * 400,
                         "pcg_iters_per_s": 0.88, "host_cpu": "AMD EPYC 9575F 64-Core Processor", "gpu_over_cpu": 278.0,
                         "C1_poisson2d_100": filler, "C2_poisson2d_4096": filler, "C3_poisson3d_512": filler},
        "cpu_baseline_reference_pcg": {"value": 1.02, "kind": "reference", "iterates_match_port": True},
        "published_table": {"ok": True, "rows": [{"n": 100, "iter": 225, "gpu_total_s": 0.01, "ref_total_s": 0.05}]},
    }
    if world > 1:
        full.update({"rccl_ranks": world, "backend": "nccl", "stage": "torch_rccl_ranks", "vs_n1": 7.1,
                     "parity_vs_n1": {"ok": True, "max_rel_diff": 2e-13, "tol": 1e-9, "iters": 20, "n1": filler},
                     "phases": {"px_update_ms": 0.9, "spmv_interior_ms": 1.6, "halo_exposed_ms": 0.0, "spmv_boundary_ms": 0.05,
                                "allreduce_1_ms": 0.03, "allreduce_2_ms": 0.03, "scalar_1_ms": 0.005, "scalar_2_ms": 0.005,
                                "r_update_ms": 0.5, "halo_ms": 0.06, "iteration_ms": 3.2, "overlap_frac": 1.0,
                                "allreduce_us": [30.0, 30.0], "note": "n" * 300},
                     "preflight": {"world": world, "backend": "nccl", "device_count": 8,
                                   "peer_access": [[1] * 8] * 8, "link_topology": filler},
                     "transport": {"halo": "RCCL send/recv (torch.distributed batch_isend_irecv)", "reductions": "RCCL"}})
    return full


@pytest.mark.parametrize("world", [1, 8])
def test_the_printed_line_is_compact_and_carries_the_whole_metric(world, tmp_path, capsys):
    """what reaches stdout is <= 6 KB and holds BOTH halves of BASELINE.json's metric where the driver
    keeps them (inside `roofline`); everything bulky is in the side file the line names"""
    import json
    import sys
    import bench_line
    full = _full_record(world)
    if world > 1:
        full["predicted"] = bench_line.predicted_iteration(1 << 30, world, 1024 * 1024, 1e3 / 40.3)
        full["predicted"]["missed_budget"] = bench_line.judge_phases(full["phases"], full["predicted"])
    side = str(tmp_path / "side.json")
    line = bench_line.emit(full, sys.stdout, side)
    text = capsys.readouterr().out.strip()
    assert text.count("\n") == 0 and len(text) <= bench_line.LINE_LIMIT and "dropped_for_length" not in line
    got = json.loads(text)
    for k in bench_line.CONTRACT_KEYS + ("config", "roofline", "cpu_baseline"):
        assert k in got, k
    r = got["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):  # the contract's roofline object
        assert k in r, k
    for k in ("pcg_iters_per_s", "pcg_frac_own_bytes", "pcg_bytes_per_iter", "pcg_loop", "csr_literal",
              "frac_8d_of_timed_kernel", "strong_n1_iters_per_s"):  # round 6: the other half of the metric
        assert k in r, k
    assert r["csr_literal"] == {"kernel": "csr_spmv_w6", "avg_launch_ms": 2.41, "frac_8d": 0.723}
    assert "note" not in r and "traffic_counters" not in r and len(r["traffic_source"]) <= 140
    c = got["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(c) and "C3_poisson3d_512" not in c
    assert c["reference_pcg_iters_per_s"] == 1.02 and len(c["sample"]) <= 200
    assert got["parity_check"] == {"ok": True, "cases": ["C3_poisson3d_512"],
                                   "worst_x_rel_diff_vs_compiled_reference": 1.3e-13}
    assert got["published_table"]["rows"] == [[100, 225, 0.01, 0.05]]
    for k in ("kernels_same_operator", "solvers", "placement_sweep", "process_mode", "strong_n1", "gpu_clocks_under_load"):
        assert k not in got and k in got["side_keys"]
    line2, back = bench_line.read(text)
    assert back == full and line2 == got
    if world > 1:
        assert got["vs_n1"] == 7.1 and got["parity_vs_n1"] == {"ok": True, "max_rel_diff": 2e-13, "tol": 1e-9, "iters": 20}
        assert "note" not in got["phases"] and got["phases"]["iteration_ms"] == 3.2
        assert got["preflight"] == {"world": world, "backend": "nccl", "device_count": 8, "peer_access_all": True}
        p = got["predicted"]
        assert abs(p["compute_ms"] - 1e3 / 40.3 / 8) < 1e-9 and 7.5 < p["vs_n1"] < 8.0
        assert p["missed_budget"] == {}  # the fake phases are on model


def test_predicted_iteration_is_design_section_5s_table_and_names_what_missed_it():
    import bench_line
    t1 = 25.0  # ms per iteration at 1024^3 on one GPU (24.9-26.6 measured)
    for world, lo, hi in ((2, 1.95, 2.0), (4, 3.85, 4.0), (8, 7.5, 7.95)):
        p = bench_line.predicted_iteration(1 << 30, world, 1 << 20, t1)
        assert lo < p["vs_n1"] < hi, (world, p["vs_n1"])
        assert abs(p["halo_ms"] - 8.0 * (1 << 20) / 153e9 * 1e3) < 1e-12 and p["halo_exposed_ms"] == 0.0
    p8 = bench_line.predicted_iteration(1 << 30, 8, 1 << 20, t1)
    assert 3.1 < p8["iteration_ms"] < 3.3 and 300 < p8["pcg_iters_per_s"] < 320
    # no one-GPU figure in the job: the byte model at the streaming rate
    pm = bench_line.predicted_iteration(1 << 30, 8, 1 << 20, None)
    assert abs(pm["compute_ms"] - (1 << 27) * 138.0 / 6.0e12 * 1e3) < 1e-9 and "vs_n1" not in pm
    slow = {"px_update_ms": p8["px_update_ms"], "spmv_interior_ms": p8["spmv_ms"], "spmv_boundary_ms": 0.0,
            "r_update_ms": p8["r_update_ms"], "halo_exposed_ms": 0.4, "halo_ms": 0.3, "allreduce_1_ms": 0.25,
            "allreduce_2_ms": 0.03, "iteration_ms": 4.1}
    missed = bench_line.judge_phases(slow, p8)
    assert set(missed) == {"halo_exposed", "halo", "allreduce_1", "iteration"} and missed["allreduce_1"] == [0.25, 0.03]


def test_a_line_that_would_be_too_long_says_what_it_dropped(tmp_path, capsys):
    import json
    import sys
    import bench_line
    full = _full_record(8)
    full["phases"] = {("phase_%d_ms" % k): 0.123456 for k in range(400)}
    bench_line.emit(full, sys.stdout, str(tmp_path / "s.json"))
    text = capsys.readouterr().out.strip()
    got = json.loads(text)
    assert len(text) <= bench_line.LINE_LIMIT and "phases" in got["dropped_for_length"] and "roofline" in got
