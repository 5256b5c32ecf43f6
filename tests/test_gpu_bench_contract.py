"""GPU: bench.py prints ONE JSON line with the driver's contract (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config) plus the
`roofline` and `cpu_baseline` objects, on a small grid so that it runs in seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True,
                         text=True, check=True, cwd=ROOT).stdout
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_json_contract_small_grid():
    d = run_bench("--grid", "96,96,96", "--steps", "5", "--warmup", "2", "--pcg-iters", "8", "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "GB/s" and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and "workload" in d["config"]
    n, nnz = d["config"]["n"], d["config"]["nnz"]
    assert n == 96 ** 3 and nnz == 7 * n - 6 * 96 * 96
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert r["kernel"] == "csr_spmv_w4" and r["algorithmic_bytes_per_launch"] == 12 * nnz + 20 * n + 4
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["dram_model_bytes_per_launch"] < r["algorithmic_bytes_per_launch"]  # no column indices are read
    assert d["roofline_general_csr"]["kernel"] == "csr_spmv_w3"
    assert d["sss_mat"]["kernel"] == "sss_spmv_w4"
    assert d["pcg_check"]["info"] == -1 and d["pcg_check"]["iter"] == 9  # tol = 0: exactly 8 iterations
    assert d["pcg_iters_per_s"] > 0 and d["value"] > 0


def test_bench_cpu_baseline_object_small_sample(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    c = bench.cpu_baseline(sample_n=48, spmv_reps=2, pcg_iters=3)
    assert c["kind"] == "port" and c["cores"] == 1 and c["unit"] == "GB/s" and c["value"] > 0
    assert "48^3" in c["sample"] and c["pcg_iters_per_s"] > 0
