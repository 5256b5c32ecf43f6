"""GPU: the single-kernel PCG loop for mid-size offset-structured systems (pysparse_amd/csrc/psp_mid.hip; VERDICT r4
"Next" #2): between 2^18 and 2^20 unknowns the whole loop of pysparse/itsolvers/src/pcg.c:91-166 runs in one cooperative
kernel -- vectors in registers, the direction vector exchanged through LDS, two grid barriers per iteration.

It must give the launch-per-phase loops' bits (same per-thread sums, same wave trees, same order of the partial sums --
for p.q the order of csr_spmv_w4's workgroups, XCD-stripe remap included): every case below runs in this process (single
kernel) and in a child with PSP_MID=0 (launch per phase) and compares info / iter / relres / x / history for EQUALITY:
2-D and 3-D grids, one and two row pairs per thread, sizes that end in the middle of a span, no preconditioner / Jacobi
with a constant diagonal / Jacobi with a varying diagonal (a 5-offset operator with random coefficients), converged runs
and every small truncation point.  Refusals (PSP_COOP_FAIL, a capacity of 4 workgroups) fall back and give the same."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json, ctypes as C, numpy as np
sys.path.insert(0, %r)
from pysparse_amd import device as dev, _capi
L = _capi.lib()
spec = json.loads(sys.argv[1])
out = []
for case in spec:
    kind, grid = case["kind"], tuple(case["grid"])
    if kind == "poisson":
        A = dev.DeviceCSR.poisson(*grid)
    elif kind == "poisson_sss":  # the same operator as an sss_mat (examples/poisson_test.py: S = L.to_sss())
        A = dev.DeviceSSS.poisson(*grid)
    elif kind == "nine":  # 9-point operator (bilinear elements on a grid): random symmetric couplings, dominant diagonal
        nx, ny = grid[0], grid[1]
        n = nx * ny
        g = np.random.default_rng(case["seed"])
        import scipy.sparse as sp
        ix = np.arange(n) %% nx
        diags, offs = [], []
        for o, ok in ((1, ix[:-1] < nx - 1), (nx - 1, ix[:n - nx + 1] > 0), (nx, np.ones(n - nx, bool)), (nx + 1, ix[:n - nx - 1] < nx - 1)):
            e = -(0.1 + g.random(n - o)) * ok
            diags += [e, e]; offs += [o, -o]
        S = sp.diags(diags, offs, shape=(n, n), format="csr")
        S = (S + sp.diags(-np.asarray(S.sum(axis=1)).ravel() + 0.3 + g.random(n))).tocsr()
        S.eliminate_zeros(); S.sort_indices()
        A = dev.DeviceCSR.from_arrays(S.shape, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    elif kind == "signs":  # diag(+1, -1, +1, ...): one offset; with b = ones p.Ap = 0 in the first iteration (pcg.c:118-120)
        n = grid[0]
        d = np.where(np.arange(n) %% 2 == 0, 1.0, -1.0)
        A = dev.DeviceCSR.from_arrays((n, n), np.arange(n + 1, dtype=np.int32), np.arange(n, dtype=np.int32), d)
    else:  # 5-offset operator with random coefficients and a varying, dominant diagonal (symmetric)
        nx, ny = grid[0], grid[1]
        n = nx * ny
        g = np.random.default_rng(case["seed"])
        import scipy.sparse as sp
        e1 = -(0.2 + g.random(n - 1)); e1[np.arange(1, n) %% nx == 0] = 0.0
        e2 = -(0.2 + g.random(n - nx))
        S = sp.diags([e2, e1, e1, e2], [-nx, -1, 1, nx], shape=(n, n), format="csr")
        S = (S + sp.diags(-np.asarray(S.sum(axis=1)).ravel() + 0.5 + g.random(n))).tocsr()
        S.sort_indices()
        A = dev.DeviceCSR.from_arrays(S.shape, S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data)
    n = A.shape[0]
    assert A.kernel_info()[0] == ("sss_spmv_w4" if kind == "poisson_sss" else "csr_spmv_w4"), A.kernel_info()
    b = np.random.default_rng(case.get("bseed", 1)).standard_normal(n)
    if case.get("b") == "ones":
        b = np.ones(n)
    if case.get("b") == "A*ones":  # the exact solution is representable: PCG ends by stagnation (pcg.c:159-162)
        b = np.empty(n); A.matvec(np.ones(n), b)
    s0 = C.c_longlong(); f0 = C.c_longlong()
    L.psp_debug_mid_count(C.byref(s0), C.byref(f0))
    for Kname in case["K"]:
      K = None if Kname == "none" else dev.DeviceJacobi(A)
      for solver in case.get("solvers", ["pcg"]):
        for tol, maxit in case["runs"]:
            x = np.zeros(n)
            r = getattr(dev, solver)(A, b, x, tol, maxit, K, hist=True)
            h = np.asarray(r[3], dtype=np.float64)
            out.append([r[0], r[1], float(r[2]).hex(), x.tobytes().hex()[:256], float(np.abs(x).sum()).hex(),
                        float(np.nansum(h)).hex(), int(np.isnan(h).sum())])
    s1 = C.c_longlong(); f1 = C.c_longlong()
    L.psp_debug_mid_count(C.byref(s1), C.byref(f1))
    out.append(["mid_solves", s1.value - s0.value, f1.value - f0.value])
print(json.dumps(out))
""" % ROOT

RUNS = [[0.0, k] for k in (1, 2, 3, 7, 16, 17, 40)] + [[1e-9, 5000]]


def _run(spec, env=None):
    e = dict(os.environ)
    e.pop("PSP_TUNING", None)
    if env:
        e.update(env, PSP_TUNING="1")
    p = subprocess.run([sys.executable, "-c", CHILD, json.dumps(spec)], env=e, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    return json.loads(p.stdout.strip().splitlines()[-1])


def _counts(out, ncases):
    return [row for row in out if row and row[0] == "mid_solves"]


PCG_CASES = [
    {"kind": "poisson", "grid": [600, 600, 0], "K": ["none", "jacobi"], "runs": RUNS},           # one layer, 176 workgroups
    {"kind": "poisson", "grid": [1024, 1024, 0], "K": ["none", "jacobi"], "runs": RUNS},         # 4096 rows per workgroup
    {"kind": "poisson", "grid": [601, 733, 0], "K": ["jacobi"], "runs": RUNS},                   # ends inside a span and a row pair
    {"kind": "poisson", "grid": [40, 40, 300], "K": ["none", "jacobi"], "runs": RUNS},           # 7 offsets, halo of 1602 rows
    {"kind": "random5", "grid": [640, 700, 0], "seed": 4, "K": ["none", "jacobi"], "runs": RUNS},  # varying diagonal: dinv array
    {"kind": "random5", "grid": [1000, 1000, 0], "seed": 5, "K": ["jacobi"], "runs": RUNS[:4] + RUNS[-1:]},
    {"kind": "poisson", "grid": [20, 20, 2500], "K": ["none", "jacobi"], "runs": RUNS},          # 7 offsets, 4096 rows per workgroup
]


def _compare_with_launch_per_phase(spec):
    """every case of the spec in one child with the single-kernel loops and in one without: all fields for equality, every
    solve of the first child one kernel, the last run of every case converged"""
    mid = _run(spec)
    ref = _run(spec, {"PSP_MID": "0"})
    want = [["mid_solves", len(c["K"]) * len(c["runs"]) * len(c.get("solvers", ["pcg"])), 0] for c in spec]
    assert [r for r in mid if r[0] == "mid_solves"] == want, [r for r in mid if r[0] == "mid_solves"]
    assert [r for r in ref if r[0] == "mid_solves"] == [["mid_solves", 0, 0]] * len(spec)
    a = [r for r in mid if r[0] != "mid_solves"]
    b = [r for r in ref if r[0] != "mid_solves"]
    assert len(a) == len(b) == sum(w[1] for w in want)
    for k, (ra, rb) in enumerate(zip(a, b)):
        assert ra == rb, (k, ra[:3], rb[:3])
    pos = 0
    for w in want:
        pos += w[1]
        assert a[pos - 1][0] == 0  # the last run of every case converges


@pytest.mark.parametrize("part", [0, 1], ids=["2d", "3d_and_varying"])
def test_single_kernel_loop_has_the_launch_per_phase_bits(part):
    _compare_with_launch_per_phase(PCG_CASES[:3] if part == 0 else PCG_CASES[3:])


@pytest.mark.parametrize("part", [0, 1], ids=["2d", "3d_and_varying"])
def test_single_kernel_minres_loop_has_the_launch_per_phase_bits(part):
    """minres.c:96-193 in one kernel (minres_mid_kernel): the same comparison, every field for equality"""
    cases = [dict(c, solvers=["minres"]) for c in (PCG_CASES[:3] if part == 0 else PCG_CASES[3:])]
    _compare_with_launch_per_phase(cases)


def test_sss_operands_take_the_single_kernel_loops_with_their_own_bits():
    """an sss_mat's product adds a row's lower entries, its diagonal and its mirrored entries in ascending column order
    (sss_mat.c:45-55): the offset table of the handle's full mirror is the same sum -- PCG and MINRES on the sss form run
    as single kernels and give the bits of the launch-per-phase loops around sss_spmv_w4"""
    spec = [{"kind": "poisson_sss", "grid": [600, 600, 0], "K": ["none", "jacobi"], "runs": RUNS, "solvers": ["pcg", "minres"]},
            {"kind": "poisson_sss", "grid": [1024, 1000, 0], "K": ["jacobi"], "runs": RUNS[:5], "solvers": ["pcg", "minres"]},
            {"kind": "poisson_sss", "grid": [40, 40, 300], "K": ["jacobi"], "runs": RUNS[:5] + RUNS[-1:], "solvers": ["pcg", "minres"]},
            {"kind": "poisson_sss", "grid": [300, 300, 0], "K": ["jacobi"], "runs": RUNS[:5] + RUNS[-1:], "solvers": ["pcg", "minres"]}]
    mid = _run(spec)
    ref = _run(spec, {"PSP_MID": "0", "PSP_COOP": "0"})
    a = [r for r in mid if r[0] != "mid_solves"]
    b = [r for r in ref if r[0] != "mid_solves"]
    assert a == b and len(a) > 0
    assert all(r[1] > 0 and r[2] == 0 for r in mid if r[0] == "mid_solves")
    assert all(r[1] == 0 for r in ref if r[0] == "mid_solves")


def test_nine_point_operators():
    """9 offsets (round 5, late): blocks of 2048 rows keep them in registers without scratch memory (512 threads), blocks of
    4096 rows with some"""
    spec = [{"kind": "nine", "grid": [70, 60, 0], "seed": 3, "K": ["jacobi"], "runs": RUNS, "solvers": ["pcg", "minres"]},  # 3 workgroups
            {"kind": "nine", "grid": [400, 300, 0], "seed": 1, "K": ["none", "jacobi"], "runs": RUNS, "solvers": ["pcg", "minres"]},
            {"kind": "nine", "grid": [1000, 900, 0], "seed": 2, "K": ["jacobi"], "runs": RUNS[:5] + RUNS[-1:], "solvers": ["pcg", "minres"]}]
    mid = _run(spec)
    ref = _run(spec, {"PSP_MID": "0", "PSP_COOP": "0"})
    a = [r for r in mid if r[0] != "mid_solves"]
    b = [r for r in ref if r[0] != "mid_solves"]
    assert a == b and len(a) > 0
    assert all(r[1] > 0 and r[2] == 0 for r in mid if r[0] == "mid_solves")
    assert a[-1][0] == 0


def test_blocks_of_8192_rows_for_constant_coefficient_pcg():
    """2^20 < n <= 2^21 (round 5, late): constant-coefficient operators run PCG in blocks of 8192 rows (no matrix registers,
    the partial sums staged in the window) -- the launch-per-phase loops' bits; MINRES and varying coefficients keep those loops"""
    spec = [{"kind": "poisson", "grid": [1200, 1200, 0], "K": ["none", "jacobi"], "runs": RUNS[:6], "solvers": ["pcg"]},
            {"kind": "poisson", "grid": [1448, 1447, 0], "K": ["jacobi"], "runs": RUNS[:5], "solvers": ["pcg"]},
            {"kind": "poisson_sss", "grid": [1100, 1300, 0], "K": ["jacobi"], "runs": RUNS[:5], "solvers": ["pcg"]},
            {"kind": "poisson", "grid": [30, 30, 1500], "K": ["jacobi"], "runs": RUNS[:5], "solvers": ["pcg"]}]
    mid = _run(spec)
    ref = _run(spec, {"PSP_MID": "0"})
    a = [r for r in mid if r[0] != "mid_solves"]
    b = [r for r in ref if r[0] != "mid_solves"]
    assert a == b and len(a) > 0
    assert all(r[1] > 0 and r[2] == 0 for r in mid if r[0] == "mid_solves")
    other = [{"kind": "poisson", "grid": [1200, 1200, 0], "K": ["jacobi"], "runs": RUNS[:2], "solvers": ["minres"]},
             {"kind": "random5", "grid": [1200, 1100, 0], "seed": 2, "K": ["jacobi"], "runs": RUNS[:2], "solvers": ["pcg"]}]
    assert all(r[1] == 0 for r in _run(other) if r[0] == "mid_solves")


def test_stagnation_exit_at_the_same_iteration():
    """pcg.c:124-139, :159-162: the kernel decides "1 + max |alpha p / x| == 1" row by row without the division
    (mid_row_moves) -- the exit must come at the launch-per-phase loops' iteration, with their bits"""
    spec = [{"kind": "poisson", "grid": [400, 400, 0], "K": ["jacobi", "none"], "runs": [[0.0, 5000]], "b": "A*ones"},
            {"kind": "poisson", "grid": [1024, 1024, 0], "K": ["jacobi"], "runs": [[0.0, 9000]], "b": "A*ones"},
            {"kind": "random5", "grid": [300, 300, 0], "seed": 9, "K": ["jacobi"], "runs": [[0.0, 5000]], "b": "A*ones"}]
    mid = _run(spec)
    ref = _run(spec, {"PSP_MID": "0", "PSP_COOP": "0"})
    a = [r for r in mid if r[0] != "mid_solves"]
    b = [r for r in ref if r[0] != "mid_solves"]
    assert a == b
    assert [r[0] for r in a] == [-5] * len(a), [r[:2] for r in a]
    assert all(r[1] > 0 and r[2] == 0 for r in mid if r[0] == "mid_solves")


def test_breakdown_exit_inside_the_kernel():
    """p.Ap == 0 (pcg.c:118-120: flag -6, x untouched by that iteration) decided inside the single kernel, at the
    launch-per-phase loops' iteration and with their x; MINRES on the same indefinite operator runs to its own end"""
    spec = [{"kind": "signs", "grid": [65536, 1, 0], "K": ["none"], "runs": [[1e-8, 50]], "b": "ones", "solvers": ["pcg", "minres"]},
            {"kind": "signs", "grid": [300000, 1, 0], "K": ["none"], "runs": [[1e-8, 50]], "b": "ones", "solvers": ["pcg"]}]
    mid = _run(spec)
    ref = _run(spec, {"PSP_MID": "0", "PSP_COOP": "0"})
    a = [r for r in mid if r[0] != "mid_solves"]
    b = [r for r in ref if r[0] != "mid_solves"]
    assert a == b
    assert a[0][:2] == [-6, 1] and a[2][:2] == [-6, 1], [r[:2] for r in a]
    assert all(r[1] > 0 and r[2] == 0 for r in mid if r[0] == "mid_solves")


@pytest.mark.parametrize("blk", ["512", "1024"])
def test_both_workgroup_sizes_give_the_same_bits(blk):
    """PSP_MID_BLK: 512 or 1024 threads per workgroup (the default picks per shape) -- the same sums in the same order"""
    spec = [{"kind": "poisson", "grid": [600, 600, 0], "K": ["jacobi"], "runs": RUNS[:5] + RUNS[-1:], "solvers": ["pcg", "minres"]},
            {"kind": "poisson", "grid": [1024, 1000, 0], "K": ["jacobi"], "runs": RUNS[:5], "solvers": ["pcg", "minres"]},
            {"kind": "poisson", "grid": [40, 40, 300], "K": ["none"], "runs": RUNS[:5], "solvers": ["pcg", "minres"]}]
    got = _run(spec, {"PSP_MID_BLK": blk})
    ref = _run(spec, {"PSP_MID": "0"})
    assert [r for r in got if r[0] != "mid_solves"] == [r for r in ref if r[0] != "mid_solves"]
    assert all(r[1] > 0 and r[2] == 0 for r in got if r[0] == "mid_solves")


def test_refused_or_failed_launch_falls_back_with_the_same_result():
    spec = [{"kind": "poisson", "grid": [600, 600, 0], "K": ["jacobi"], "runs": [[0.0, 9], [1e-9, 5000]],
             "solvers": ["pcg", "minres"]}]
    want = [r for r in _run(spec, {"PSP_MID": "0"}) if r[0] != "mid_solves"]
    for env, fb in (({"PSP_COOP_FAIL": "1"}, 4), ({"PSP_COOP_CAPACITY": "4"}, 0)):
        got = _run(spec, env)
        assert [r for r in got if r[0] != "mid_solves"] == want, env
        assert [r for r in got if r[0] == "mid_solves"] == [["mid_solves", 0, fb]], (env, got[-1])
