"""CPU (no GPU needed): BASELINE.json configs[0] -- "2D Poisson 5-pt 100x100 CSR SpMV + Jacobi-PCG on CPU
(examples/demo_pcg.py plumbing, no GPU)" -- through the library's OPT-IN host mode, PSP_DEVICE=cpu
(pysparse_amd/csrc/psp_cpu.hip).  The mode is never selected implicitly: tests/test_capi_symbols.py and
tests/test_spmatrix_host.py keep asserting that without the variable a GPU-less process fails with "no HIP device"."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([sys.executable] + args, env=env, capture_output=True, text=True, cwd=ROOT, timeout=900)


def test_host_mode_runs_the_drop_in_modules_against_oracle_and_goldens(golden_dir):
    p = run([os.path.join(ROOT, "tests", "cpu_mode_child.py")], {"PSP_DEVICE": "cpu"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["cases"] >= 40
    with open(os.path.join(golden_dir, "ref_pcg.json")) as f:
        g1 = json.load(f)["G1_jacobi"]
    info, it, relres, err = out["G1"]
    assert (info, it) == (g1["info"], g1["iter"]) == (0, 160)  # iter 160, relres 8.8679e-07
    assert abs(relres - g1["relres"]) <= 5e-3 * g1["relres"] and abs(relres - 8.8679e-07) < 1e-10
    assert abs(err - g1["err_inf"]) <= 1e-9


def test_demo_pcg_script_on_a_machine_without_gpu():
    p = run([os.path.join(ROOT, "examples", "demo_pcg.py"), "--poisson", "100"], {"PSP_DEVICE": "cpu"})
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    rows = [l.split() for l in p.stdout.splitlines() if "e-0" in l]
    assert len(rows) == 3  # no preconditioner, jacobi, ssor
    assert rows[0][3] == "160" and rows[1][0] == "160" and rows[0][4] == "8.9e-07" and rows[0][6] == "0"
    assert int(rows[2][0]) < 160  # ssor: fewer iterations (demo_pcg.py's third column)


def test_host_mode_is_opt_in_only():
    """without the variable nothing changes: no GPU -> "no HIP device"; a misspelt value is not the mode"""
    import pytest
    sys.path.insert(0, ROOT)
    from pysparse_amd import device
    if device.device_count() > 0:
        pytest.skip("a GPU is present")
    code = ("import sys; sys.path.insert(0, %r)\nfrom pysparse.sparse import spmatrix\n"
            "try:\n spmatrix.poisson_csr(4, 4)\n print('built')\nexcept RuntimeError as e:\n print('error:', e)\n") % ROOT
    for env in ({}, {"PSP_DEVICE": "CPU"}, {"PSP_DEVICE": "host"}, {"PSP_TUNING": "1"}):
        e = {k: v for k, v in os.environ.items() if k != "PSP_DEVICE"}
        e.update(env)
        p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert "no HIP device" in p.stdout and "built" not in p.stdout, (env, p.stdout, p.stderr[-500:])
