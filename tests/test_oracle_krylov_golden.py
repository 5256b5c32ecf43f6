"""CPU: the oracle's restatements of pcg / minres / cgs / bicgstab / qmrs / gmres (oracle/pysparse_oracle.c) against
  - tests/golden/ref_krylov.json: what the reference's OWN kernels -- pysparse/itsolvers/src/{pcg,minres,cgs,bicgstab,
    qmrs,gmres}.c compiled unmodified into oracle/_ref/libref_krylov.so (oracle/Makefile, oracle/ref_krylov_harness.c) --
    return on the cases of tests/krylov_cases.py (always), and
  - those compiled kernels themselves, live, when oracle/_ref is present (this container; the GPU box gets the prebuilt file)."""
import json
import os

import numpy as np
import pytest

import krylov_cases as KC


@pytest.fixture(scope="module")
def gold(golden_dir):
    with open(os.path.join(golden_dir, "ref_krylov.json")) as f:
        return json.load(f)["cases"], np.load(os.path.join(golden_dir, "ref_krylov_iterates.npz"))


def test_golden_file_covers_every_case(gold):
    cases, _ = gold
    assert set(cases) == set(KC.CASES)
    for name, rec in cases.items():
        assert rec["case"] == json.loads(json.dumps(KC.CASES[name])), name  # the fixture was made from these inputs
    solvers = {c["solver"] for c in KC.CASES.values()}
    assert solvers == {"pcg", "minres", "cgs", "bicgstab", "qmrs", "gmres"}


@pytest.mark.parametrize("name", sorted(KC.CASES))
def test_oracle_restatement_matches_reference_golden(oracle, gold, name):
    cases, its = gold
    KC.check_against_golden(name, KC.run_oracle(oracle, KC.CASES[name]), cases[name]["expect"], its)


def test_compiled_reference_reproduces_its_goldens_live(oracle, gold):
    if not oracle.have_ref_krylov():
        pytest.skip("oracle/_ref/libref_krylov.so not built (needs /root/reference)")
    cases, its = gold
    for name, case in KC.CASES.items():
        info, it, rr, x, rc = KC.run_reference(oracle, case)
        KC.check_against_golden(name, (info, it, rr, x), cases[name]["expect"], its)
        assert rc == cases[name]["expect"]["rc"], name


def test_oracle_matches_compiled_reference_on_fresh_problems(oracle):
    """beyond the committed cases: random right-hand sides / start vectors, every solver, None / Jacobi / 2-step Jacobi,
    CSR and SSS operators -- restatement against the compiled kernel after 8 iterations and (pcg, minres, qmrs, gmres)
    at convergence"""
    if not oracle.have_ref_krylov():
        pytest.skip("oracle/_ref/libref_krylov.so not built (needs /root/reference)")
    rng = np.random.default_rng(11)
    mats = [oracle.poisson_csr(30, 20), oracle.poisson_sss(25, 31), oracle.poisson_csr(9, 8, 7), KC.nonsym_csr(oracle, 500, 2)]
    for mi, A in enumerate(mats):
        n = A.shape[0]
        d = A.diag if isinstance(A, oracle.SSS) else A.diagonal()
        b = rng.standard_normal(n)
        for solver in ("pcg", "minres", "cgs", "bicgstab", "qmrs", "gmres"):
            if mi == 3 and solver in ("pcg", "minres", "qmrs"):
                continue  # the symmetric solvers on a non-symmetric matrix prove nothing
            for K in (None, ("jacobi", oracle.jacobi_dinv(d), 1), ("jacobi", oracle.jacobi_dinv(d, 0.9), 2)):
                for tol, maxit in ((0.0 if solver in ("pcg", "minres") else 1e-300, 8), (1e-10, 2000)):
                    x0 = rng.standard_normal(n)
                    x1, x2 = x0.copy(), x0.copy()
                    r1 = oracle.solve(solver, A, b, x1, tol, maxit, K, dim=12)
                    r2 = oracle.ref_krylov(solver, A, b, x2, tol, maxit, K, dim=12)
                    assert r1[0] == r2[0], (mi, solver, r1, r2)
                    scale = np.abs(x2).max()
                    if maxit == 8 or solver in ("pcg", "minres", "qmrs", "gmres"):
                        assert r1[1] == r2[1], (mi, solver, r1, r2)
                        assert np.abs(x1 - x2).max() <= (1e-12 if maxit == 8 else 1e-10) * scale, (mi, solver, K is None)
                    else:
                        assert abs(r1[1] - r2[1]) <= max(3, 0.4 * r2[1]) and np.abs(x1 - x2).max() <= 1e-6 * scale


def test_compiled_reference_callback_failure_returns_minus_one(oracle):
    """a raising matvec / precon makes every kernel return -1 at once (SpMatrix_MATVEC / SpMatrix_PRECON macros,
    e.g. pcg.c:8-11, minres.c:38-41); the restatement reports the same event as -100"""
    if not oracle.have_ref_krylov():
        pytest.skip("oracle/_ref/libref_krylov.so not built (needs /root/reference)")
    A = oracle.poisson_csr(10, 10)
    b = np.ones(100)
    dinv = oracle.jacobi_dinv(A.diagonal())
    for solver in ("pcg", "minres", "cgs", "bicgstab", "qmrs", "gmres"):
        assert oracle.ref_krylov(solver, A, b, np.zeros(100), 1e-10, 50, mv_fail_after=2)[3] == -1
        assert oracle.ref_krylov(solver, A, b, np.zeros(100), 1e-10, 50, ("jacobi", dinv), pc_fail_after=1)[3] == -1
