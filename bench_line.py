"""bench_line.py -- the ONE JSON line bench.py prints, kept small enough to survive whoever stores it (round 6).

bench.py assembles a full record (every leg, every note: 15-20 KB).  The driver that runs it keeps the parsed contract
keys plus `roofline` / `cpu_baseline`, and only a tail of the raw line -- so half of BASELINE.json's metric ("... + PCG
iters/sec") never reached the record.  Now:

  * `compact(full)` is what goes to stdout: the contract keys, `roofline` (with the PCG figure, the literal-CSR kernel's
    fraction in SURVEY 8d's bytes and the 1024^3 one-GPU figure INSIDE it), a five-key `cpu_baseline`, parity verdicts,
    and for N > 1 `phases` / `predicted` / `vs_n1` -- at most LINE_LIMIT characters (tests/test_bench_region.py);
  * the full record goes to a side file (`side_file` in the line; default gpurun_out/bench_side_n<N>.json under the
    repo, --side-file to choose) -- `read()` gives both back to tests and tools.

Nothing here touches the GPU, measures anything, or imports the oracle."""
import json
import os
import tempfile

from bench_common import HBM_PEAK_GBPS, ROOT

LINE_LIMIT = 6000  # characters of the printed line (<= 6 KB)

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")

# streaming rate the library's vector kernels and index-free products reach on MI355X (5.95-6.27
# TB/s measured over five rounds); the per-phase budgets of `predicted` are bytes / this
STREAM_RATE_BPS = 6.0e12
XGMI_LINK_BPS = 153e9        # one xGMI link (MI355X_MICROARCH.md): a halo plane goes to ONE neighbour over ONE link
ALLREDUCE_US = 30.0          # 16-byte all-reduce over 2-8 ranks: 20-40 us (latency-bound), not hidden
SCALAR_STEP_US = 5.0         # the one-block scalar step behind each reduction (a dependent launch)


def default_side_path(n_gpus, stage=""):
    name = "bench_side_n%d%s.json" % (n_gpus, ("_" + stage) if stage else "")
    return os.path.join(ROOT, "gpurun_out", name)


def _write_side(full, path):
    """write the full record; falls back to the temp directory when the repo is read-only.  Returns the path written
    (relative to the repo when inside it) or None."""
    for p in (path, os.path.join(tempfile.gettempdir(), os.path.basename(path))):
        try:
            os.makedirs(os.path.dirname(p), exist_ok=True)
            tmp = p + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump(full, f)
            os.replace(tmp, p)
            return os.path.relpath(p, ROOT) if os.path.abspath(p).startswith(ROOT + os.sep) else p
        except OSError:
            continue
    return None


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d}


def predicted_iteration(n_total, world, nplane, t1_iter_ms=None, kbytes_row=74.0, vec_bytes_row=64.0):
    """DESIGN.md section 5's prediction for ONE Jacobi-PCG iteration of the row-partitioned loop at `world` ranks, phase by
    phase, so that a measured `phases` object can be read against it at a glance.

    Model: a rank's slab of n_total / world rows moves at the one-GPU streaming rate (the path is HBM-bound and every
    kernel is the one-GPU kernel on fewer rows); a halo plane of `nplane` doubles goes to one neighbour over one xGMI
    link and is hidden behind the interior rows (halo_exposed 0, overlap 1); the two packed 16-byte all-reduces and the
    scalar steps behind them are NOT hidden.  t1_iter_ms: the one-GPU iteration of the same problem measured in the same
    job (`strong_n1`), when there is one -- then `compute_ms` is t1 / world instead of the byte model."""
    rows = n_total / float(world)
    spmv_ms = rows * kbytes_row / STREAM_RATE_BPS * 1e3
    px_ms = rows * 40.0 / STREAM_RATE_BPS * 1e3      # r, p, x read; p, x written
    r_ms = rows * 24.0 / STREAM_RATE_BPS * 1e3       # q, r read; r written
    model_ms = rows * (kbytes_row + vec_bytes_row) / STREAM_RATE_BPS * 1e3
    compute_ms = (t1_iter_ms / world) if t1_iter_ms else model_ms
    comm_ms = 0.0 if world == 1 else 2.0 * (ALLREDUCE_US + SCALAR_STEP_US) * 1e-3
    halo_ms = 0.0 if world == 1 else 8.0 * nplane / XGMI_LINK_BPS * 1e3
    it_ms = compute_ms + comm_ms
    out = {"px_update_ms": px_ms, "spmv_ms": spmv_ms, "r_update_ms": r_ms, "halo_ms": halo_ms, "halo_exposed_ms": 0.0,
           "overlap_frac": 1.0, "allreduce_us": [ALLREDUCE_US, ALLREDUCE_US] if world > 1 else [0.0, 0.0],
           "scalar_us": [SCALAR_STEP_US, SCALAR_STEP_US], "compute_ms": compute_ms, "iteration_ms": it_ms,
           "pcg_iters_per_s": 1e3 / it_ms, "basis": "t1/N" if t1_iter_ms else "bytes/6.0TB/s"}
    if t1_iter_ms:
        out["vs_n1"] = t1_iter_ms / it_ms
    return out


def judge_phases(phases, pred):
    """which phases of a measured iteration missed the budget `pred` gives them: {phase: [measured, budget]} for every
    phase more than 25 % (and more than 20 us) over; empty = on model"""
    if not phases or not pred:
        return None
    pairs = {
        "px_update": (phases.get("px_update_ms"), pred["px_update_ms"]),
        "spmv": ((phases.get("spmv_interior_ms") or 0.0) + (phases.get("spmv_boundary_ms") or 0.0)
                 if "spmv_interior_ms" in phases else phases.get("spmv_local_ms"), pred["spmv_ms"]),
        "r_update": (phases.get("r_update_ms"), pred["r_update_ms"]),
        "halo_exposed": (phases.get("halo_exposed_ms"), 0.02),
        "halo": (phases.get("halo_ms"), pred["halo_ms"]),
        "allreduce_1": (phases.get("allreduce_1_ms"), pred["allreduce_us"][0] * 1e-3),
        "allreduce_2": (phases.get("allreduce_2_ms"), pred["allreduce_us"][1] * 1e-3),
        "iteration": (phases.get("iteration_ms"), pred["iteration_ms"]),
    }
    missed = {}
    for name, (got, budget) in pairs.items():
        if got is None or budget is None:
            continue
        if got > 1.25 * budget and got - budget > 0.02:
            missed[name] = [round(got, 4), round(budget, 4)]
    return missed


def compact(full, side_file=None):
    """the printed line: see the module docstring"""
    line = {k: full.get(k) for k in CONTRACT_KEYS}
    cfg = full.get("config") or {}
    line["config"] = _pick(cfg, ("workload", "n", "nnz", "rows_per_gpu", "parallelism", "scaling_mode"))
    for k in ("pcg_iters_per_s", "pct_hbm_peak", "effective_csr_model_GBps"):
        if k in full:
            line[k] = full[k]
    r = full.get("roofline")
    if r:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                "avg_launch_ms", "median_launch_ms", "csr_model_bytes_per_launch", "frac_8d_of_timed_kernel",
                "frac_8d_note", "csr_literal", "csr_model_frac_of_streaming_kernel", "streaming_kernel",
                "pcg_iters_per_s", "pcg_loop", "pcg_launches_per_iter", "pcg_bytes_per_iter", "pcg_frac_own_bytes",
                "strong_n1_iters_per_s", "strong_n1_spmv_frac", "stream_ceiling_GBps", "frac_of_stream_ceiling",
                "placement_level")
        rr = _pick(r, keep)
        ts = r.get("traffic_source")
        if ts:
            rr["traffic_source"] = ts[:140]
        line["roofline"] = rr
    cb = full.get("cpu_baseline")
    if cb:
        c = _pick(cb, ("value", "unit", "cores", "kind", "pcg_iters_per_s", "host_cpu", "gpu_over_cpu"))
        c["sample"] = (cb.get("sample") or "")[:200]
        ref = full.get("cpu_baseline_reference_pcg")
        if ref:
            c["reference_pcg_iters_per_s"] = ref.get("value")
            c["reference_iterates_match_port"] = ref.get("iterates_match_port")
        line["cpu_baseline"] = c
    if "parity_check" in full:
        pc = full["parity_check"]
        worst = 0.0
        for case in (pc.get("cases") or {}).values():
            for name in ("pcg", "minres"):
                rec = case.get(name) or {}
                for key in ("vs_reference_module_kernel", "vs_reference_pcg"):
                    if key in rec:
                        worst = max(worst, rec[key].get("x_max_rel_diff", 0.0))
        line["parity_check"] = {"ok": pc.get("ok"), "cases": sorted((pc.get("cases") or {}).keys()),
                                "worst_x_rel_diff_vs_compiled_reference": worst}
    if "published_table" in full:
        pt = full["published_table"]
        line["published_table"] = {"ok": pt.get("ok"),
                                   "rows": [[r_["n"], r_.get("iter"), r_.get("gpu_total_s"), r_.get("ref_total_s")]
                                            for r_ in pt.get("rows", [])],
                                   "columns": ["n", "iter", "gpu assembly+solve s", "compiled reference 1 core s"]}
    if "config5" in full and isinstance(full["config5"], dict):
        c5 = full["config5"]
        if "error" in c5:
            line["config5"] = {"error": str(c5["error"])[:200]}
        else:
            tts = c5.get("time_to_solution") or {}
            line["config5"] = {
                "source": str(c5.get("source", ""))[:90], "n": c5.get("n"), "nnz_lower": c5.get("nnz_lower"),
                "kernel": c5.get("kernel"), "spmv_ms": c5.get("spmv_ms"), "setup_ms": c5.get("setup_ms"),
                "sss_model_frac_of_peak": c5.get("sss_model_frac_of_peak"),
                "csr_model_frac_of_peak": c5.get("csr_model_frac_of_peak"),
                "time_to_solution": _pick(tts, ("end_to_end_ms", "upload_ms", "first_solve_ms", "second_solve_ms", "kernel")),
                "cold": _pick(c5.get("cold") or {}, ("kernel", "spmv_ms", "minres_us_per_iteration")),
                "minres": _pick(c5.get("minres") or {}, ("info", "iter", "us_per_iteration")),
                "parity_ok": (c5.get("parity") or {}).get("ok")}
    if "provenance" in full:
        line["provenance"] = _pick(full["provenance"], ("build_id", "match"))
    for k in ("rccl_ranks", "backend", "stage", "launcher_kind", "vs_n1", "dry_run", "ranks", "distinct_devices",
              "reductions"):
        if k in full and full[k] is not None:
            line[k] = full[k]
    if "transport" in full:
        line["transport"] = {k: str(v)[:80] for k, v in full["transport"].items()}
    if "parity_vs_n1" in full:
        line["parity_vs_n1"] = _pick(full["parity_vs_n1"], ("ok", "max_rel_diff", "tol", "iters", "same_info_iter"))
    if "phases" in full and full["phases"]:
        line["phases"] = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in full["phases"].items() if k != "note"}
    if "predicted" in full:
        line["predicted"] = full["predicted"]
    if "preflight" in full and full["preflight"]:
        pf = full["preflight"]
        pa = pf.get("peer_access")
        line["preflight"] = _pick(pf, ("world", "backend", "device_count"))
        if pa is not None:
            line["preflight"]["peer_access_all"] = bool(all(all(row) for row in pa))
    if "launcher" in full:
        line["launcher"] = compact_launcher(full["launcher"])
    if "error" in full:
        line["error"] = full["error"]
    if side_file:
        line["side_file"] = side_file
        line["side_keys"] = sorted(k for k in full if k not in line)
    return line


def compact_launcher(launcher):
    out = {k: v for k, v in launcher.items() if k != "fallback_from"}
    ff = []
    for f in launcher.get("fallback_from") or []:
        g = {k: (v[:200] if isinstance(v, str) else v) for k, v in f.items() if k != "stderr_tail"}
        if f.get("stderr_tail"):
            g["stderr_tail"] = [l[:200] for l in f["stderr_tail"][:4]]
        ff.append(g)
    out["fallback_from"] = ff
    return out


def emit(full, stream, side_path=None):
    """write the full record to the side file and print the compact line on `stream`.  Returns the line (dict)."""
    path = side_path or default_side_path(full.get("n_gpus") or 1, full.get("stage") or "")
    written = _write_side(full, path)
    line = compact(full, written)
    text = json.dumps(line)
    if len(text) > LINE_LIMIT:  # never silently: drop the optional objects largest first, and say so
        for k in ("side_keys", "predicted", "phases", "published_table", "preflight", "transport", "parity_check"):
            if k in line and len(text) > LINE_LIMIT:
                line.pop(k)
                line["dropped_for_length"] = line.get("dropped_for_length", []) + [k]
                text = json.dumps(line)
    print(text, file=stream, flush=True)
    return line


def attach_launcher(rec, launcher):
    """the ladder relays a stage's line: `launcher` goes into the side file in full and into the line in short"""
    side = rec.get("side_file")
    if side:
        p = side if os.path.isabs(side) else os.path.join(ROOT, side)
        try:
            full = json.load(open(p))
            full["launcher"] = launcher
            _write_side(full, p)
        except (OSError, ValueError):
            pass
    rec["launcher"] = compact_launcher(launcher)
    return rec


def read(stdout_text):
    """(line, full) from what a bench.py run printed: the last JSON line of stdout and the side file it names (full =
    the line itself when it names none: error lines)"""
    lines = [l for l in stdout_text.strip().splitlines() if l.startswith("{")]
    if not lines:
        raise ValueError("no JSON line in the bench output")
    line = json.loads(lines[-1])
    side = line.get("side_file")
    if not side:
        return line, line
    p = side if os.path.isabs(side) else os.path.join(ROOT, side)
    full = json.load(open(p))
    return line, full
