"""GPU parity: precon.ssor on an sss_mat (level-scheduled sweeps) vs the CPU oracle's sequential
sweeps (preconmodule.c:95-223).  Every row performs the reference's operations in the reference's
order, so the bar is bit equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rng_vec(n, seed=0):
    return np.random.default_rng(seed).standard_normal(n)


def random_sss(O, n, seed, max_lower=12):
    rng = np.random.default_rng(seed)
    lens = np.minimum(rng.integers(0, max_lower + 1, size=n), np.arange(n))
    ind = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(lens, out=ind[1:])
    col = np.concatenate([np.sort(rng.choice(i, size=lens[i], replace=False)) for i in range(n)] +
                         [np.zeros(0, dtype=np.int64)]).astype(np.int32)
    val = 0.1 * rng.standard_normal(ind[-1])
    diag = 4.0 + rng.random(n)  # diagonally dominant enough to keep the sweeps tame
    return O.SSS(n, val, diag, col, ind)


def cases(oracle, which):
    from pysparse_amd.device import DeviceSSS
    if which == "poisson2d":
        S = oracle.poisson_sss(37, 29)
    elif which == "poisson3d":
        S = oracle.poisson_sss(13, 11, 9)
    elif which == "tendigit":
        S = oracle.tendigit_sss(3000)
    elif which == "diagonal":  # no off-diagonal entry: one level in each direction
        S = oracle.SSS(50, np.zeros(0), 2.0 + np.arange(50.0), np.zeros(0, dtype=np.int32), np.zeros(51, dtype=np.int32))
    else:
        S = random_sss(oracle, 2500, 11)
    return S, DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)


@pytest.mark.parametrize("omega,steps", [(1.0, 1), (1.0, 2), (1.4, 1), (0.8, 3), (1.0, 0), (1.2, 0)])
@pytest.mark.parametrize("which", ["poisson2d", "poisson3d", "tendigit", "diagonal", "random"])
def test_ssor_precon_bit_exact(oracle, which, omega, steps):
    from pysparse_amd.device import DeviceSSOR
    S, D = cases(oracle, which)
    K = DeviceSSOR(D, omega, steps)
    assert K.shape == (S.n, S.n)
    x = rng_vec(S.n, 5)
    y_ref = np.full(S.n, 3.25)
    oracle.ssor_apply(S, x, y_ref, omega, steps)
    y = np.full(S.n, 3.25)
    K.precon(x, y)
    assert np.array_equal(y, y_ref)
    if which == "poisson2d":
        assert K.levels == (37 + 29 - 1, 37 + 29 - 1)  # hyperplanes i + j = const
    if which == "diagonal":
        assert K.levels == (1, 1)
    with pytest.raises(ValueError):
        K.precon(x[:-1], y)
    with pytest.raises(ValueError):
        K.precon(x, y[::2])


def grid_sss(O, nx, ny, nz, seed, keep=0.85):
    """variable-coefficient 7-point-like operator in natural ordering with some couplings removed: lower offsets
    -nx*ny, -nx, -1 (never across a grid line or plane), random values, dominant diagonal"""
    rng = np.random.default_rng(seed)
    P = O.poisson_sss(nx, ny, nz)
    sel = rng.random(P.val.size) < keep
    lens = np.add.reduceat(sel.astype(np.int64), P.ind[:-1].astype(np.int64)) * (np.diff(P.ind) > 0)
    ind = np.zeros(P.n + 1, dtype=np.int32)
    np.cumsum(lens, out=ind[1:])
    col = P.col[sel]
    val = -(0.2 + rng.random(col.size))
    diag = 6.5 + rng.random(P.n)
    return O.SSS(P.n, val, diag, col, ind)


@pytest.mark.parametrize("omega,steps", [(1.0, 1), (1.0, 3), (1.35, 1), (0.7, 2)])
@pytest.mark.parametrize("grid", [(16, 4, 8), (40, 30, 20), (33, 17, 9), (7, 5, 64), (100, 3, 3), (17, 64, 2)])
def test_ssor_grid_operators_bit_exact(oracle, grid, omega, steps):
    """variable-coefficient 3-D grid operators with missing couplings (rows of 0-3 lower entries: the padded
    slot-major form of the level-ordered triangle), thin and thick grids -- bit-identical to the sequential sweeps
    of the oracle; likewise with one extra coupling that wraps around a grid line."""
    from pysparse_amd.device import DeviceSSOR, DeviceSSS
    S = grid_sss(oracle, *grid, seed=sum(grid))
    D = DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)
    K = DeviceSSOR(D, omega, steps)
    x = rng_vec(S.n, 3)
    y_ref = np.full(S.n, -1.5)
    oracle.ssor_apply(S, x, y_ref, omega, steps)
    y = np.full(S.n, -1.5)
    K.precon(x, y)
    assert np.array_equal(y, y_ref)
    # the same matrix with ONE extra coupling that wraps around a grid line (row i = 0 of line 1 to the last point
    # of line 0): offsets unchanged, but not a grid operator any more
    nx = grid[0]
    if S.n >= 512:
        r = nx  # first point of the second grid line: its offset -1 neighbour is the end of the first line
        lo, hi = S.ind[r], S.ind[r + 1]
        if (r - 1) not in S.col[lo:hi]:
            col = np.insert(S.col, hi, r - 1)
            val = np.insert(S.val, hi, -0.25)
            ind = S.ind.copy()
            ind[r + 1:] += 1
            S2 = oracle.SSS(S.n, val, S.diag, col, ind)
            D2 = DeviceSSS.from_arrays(S2.n, S2.ind, S2.col, S2.val, S2.diag)
            y_ref = np.zeros(S.n)
            oracle.ssor_apply(S2, x, y_ref, omega, steps)
            y = np.zeros(S.n)
            DeviceSSOR(D2, omega, steps).precon(x, y)
            assert np.array_equal(y, y_ref)


def test_ssor_requires_sss(oracle):
    from pysparse_amd.device import DeviceCSR, DeviceSSOR
    with pytest.raises(TypeError):
        DeviceSSOR(DeviceCSR.poisson(5, 5))


@pytest.mark.parametrize("omega,steps", [(1.0, 1), (1.3, 1), (1.0, 2)])
@pytest.mark.parametrize("grid", [(100, 100, 0), (24, 24, 24)])
def test_pcg_with_ssor_matches_oracle(oracle, grid, omega, steps):
    """demo_pcg.py's third column: pcg(A, b, x, tol, maxit, ssor(S, omega, steps)), b = A*e"""
    from pysparse_amd.device import DeviceSSOR, DeviceSSS, pcg
    S = oracle.poisson_sss(*grid)
    D = DeviceSSS.poisson(*grid)
    n = S.n
    b = np.empty(n)
    S.matvec(np.ones(n), b)
    xo = np.zeros(n)
    info_o, it_o, rr_o, hist_o = oracle.pcg_ssor(S, b, xo, 1e-8, 2 * n, omega, steps, hist=True)
    xs = np.zeros(n)
    info, it, rr, hist = pcg(D, b, xs, 1e-8, 2 * n, DeviceSSOR(D, omega, steps), hist=True)
    assert (info, it) == (info_o, it_o) and info == 0
    assert abs(rr - rr_o) <= 1e-9 * rr_o
    assert np.allclose(hist[:it + 1], hist_o[:it + 1], rtol=1e-9, atol=0)
    assert np.abs(xs - xo).max() <= 1e-12 * np.abs(xo).max()
    # SSOR beats Jacobi on the Poisson operator (the reason for demo_pcg.py's third column)
    xj = np.zeros(n)
    _, it_j, _ = oracle.pcg(S, b, xj, 1e-8, 2 * n, oracle.jacobi_dinv(S.diag))
    assert it < it_j


def test_dropin_module_ssor(oracle):
    """pysparse.precon.precon.ssor through the extension modules, the way demo_pcg.py:84-98 uses it"""
    from pysparse.sparse import spmatrix
    from pysparse.itsolvers.krylov import pcg
    from pysparse.precon import precon
    n1 = 60
    S = spmatrix.poisson_sss(n1, n1)
    n = n1 * n1
    assert S.shape == (n, n)
    K = precon.ssor(S)  # omega = 1.0, steps = 1
    assert K.shape == (n, n)
    So = oracle.poisson_sss(n1, n1)
    x = rng_vec(n, 1)
    y, y_ref = np.zeros(n), np.zeros(n)
    K.precon(x, y)
    oracle.ssor_apply(So, x, y_ref, 1.0, 1)
    assert np.array_equal(y, y_ref)
    K2 = precon.ssor(S, 1.5, 2)
    K2.precon(x, y)
    oracle.ssor_apply(So, x, y_ref, 1.5, 2)
    assert np.array_equal(y, y_ref)
    with pytest.raises(TypeError):
        precon.ssor(spmatrix.poisson_csr(n1, n1))  # "O!" with SSSMatType (preconmodule.c:499)
    with pytest.raises(ValueError):
        K.precon(x[:-1], y)
    b = np.empty(n)
    S.matvec(np.ones(n), b)
    xs, xo = np.zeros(n), np.zeros(n)
    res = pcg(S, b, xs, 1e-6, 2 * n, K)
    ref = oracle.pcg_ssor(So, b, xo, 1e-6, 2 * n, 1.0, 1)
    assert res[:2] == ref[:2] and abs(res[2] - ref[2]) <= 1e-9 * ref[2]
    assert np.abs(xs - xo).max() <= 1e-12 * np.abs(xo).max()


def test_ssor_levels_by_relaxation_fallback():
    """The level schedule comes from Kahn's algorithm on the device; the relaxation sweeps it replaced remain as the
    fallback for dependency graphs deeper than its counter table (PSP_SSOR_KAHN=0 selects them; read once per process):
    the bit-exactness tests of this file once more in a fresh interpreter with that switch."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, PSP_TUNING="1", PSP_SSOR_KAHN="0")
    here = os.path.abspath(__file__)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", here, "-k",
                        "bit_exact and not relaxation"], env=env, cwd=os.path.dirname(os.path.dirname(here)),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]


@pytest.mark.parametrize("omega,steps", [(1.0, 1), (1.25, 2)])
@pytest.mark.parametrize("grid,keep", [((1500, 700, 0), 1.0), ((700, 1500, 0), 1.0), ((64, 64, 64), 1.0),
                                       ((128, 128, 64), 1.0), ((3000, 9, 9), 1.0), ((5000, 3, 0), 1.0),
                                       ((900, 800, 0), 0.85), ((900, 800, 0), 0.999), ((80, 70, 60), 0.9995),
                                       ((2100, 2100, 0), 1.0)])
def test_ssor_runs_of_narrow_levels_bit_exact(oracle, grid, keep, omega, steps):
    """Round 3: runs of narrow levels are walked by one workgroup that hands x from level to level through an LDS ring
    (ssor_run_kernel): levels of more than one tick (1500 / 2100 rows), whole 3-D schedules (64^3: every level <= 3072
    rows), runs at both thin ends of a schedule whose middle is launched level by level (128 x 128 x 64: the second run
    starts with dependencies computed by other kernels), long grid lines (dependencies 3000 slots back), rows with missing
    couplings (levels that reach further back than the ring split the runs).  Same bits as the oracle's sequential
    sweeps (preconmodule.c:95-193)."""
    from pysparse_amd.device import DeviceSSOR, DeviceSSS
    S = grid_sss(oracle, *grid, seed=7 + sum(grid), keep=keep)
    D = DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)
    K = DeviceSSOR(D, omega, steps)
    rf, rb, levels, slots = K.lds_runs
    if keep > 0.9 and not K.bricks:
        assert rf >= 1 and rb >= 1 and slots > 0
    if keep == 0.999:
        assert rf > 1 and levels < sum(K.levels)  # split at the levels that reach too far back
    if grid == (128, 128, 64):  # levels of up to 8192 rows: a 3-D grid operator like this one is swept brick by brick ...
        assert K.bricks == 4 * 4 * 2 and rf == 0 and rb == 0
    if grid == (80, 70, 60) and keep < 1.0:  # ... unless a dependency inside a brick lies too far back (missing couplings)
        assert K.bricks == 0
    if grid in ((1500, 700, 0), (700, 1500, 0), (64, 64, 64), (2100, 2100, 0)):
        assert levels == sum(K.levels) and slots == 2 * S.n      # the whole schedule
    x = rng_vec(S.n, 3)
    y_ref = np.full(S.n, -1.5)
    oracle.ssor_apply(S, x, y_ref, omega, steps)
    y = np.full(S.n, -1.5)
    K.precon(x, y)
    assert np.array_equal(y, y_ref)
    K.precon(x, y)  # the replayed graph
    assert np.array_equal(y, y_ref)


def test_ssor_runs_switch_is_an_ab_switch(oracle):
    """PSP_TUNING=1 PSP_SSOR_LDS=0 keeps every level on its own launch (the round-2 path); both give the oracle's bits"""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from pysparse_amd.device import DeviceSSOR, DeviceSSS\n"
        "D = DeviceSSS.poisson(300, 200)\n"
        "K = DeviceSSOR(D, 1.0, 1)\n"
        "x = np.random.default_rng(1).standard_normal(60000); y = np.zeros(60000)\n"
        "K.precon(x, y)\n"
        "print(K.lds_runs[0], repr(float(y @ y)), repr(float(y[12345])))\n" % os.path.dirname(os.path.dirname(__file__)))
    outs = []
    for env in ({}, {"PSP_TUNING": "1", "PSP_SSOR_LDS": "0"}):
        e = dict(os.environ)
        e.pop("PSP_TUNING", None)
        e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.split())
    assert int(outs[0][0]) >= 1 and int(outs[1][0]) == 0
    assert outs[0][1:] == outs[1][1:]


@pytest.mark.parametrize("omega,steps", [(1.0, 1), (1.0, 2), (1.3, 1), (0.8, 2)])
@pytest.mark.parametrize("grid,keep", [((128, 128, 64), 1.0), ((100, 90, 80), 1.0), ((65, 200, 70), 1.0),
                                       ((130, 40, 140), 1.0), ((96, 96, 96), 0.97), ((160, 150, 33), 1.0)])
def test_ssor_bricks_bit_exact(oracle, grid, keep, omega, steps):
    """Round 3: 3-D grid operators whose levels are too wide for a run (more than 4096 rows) are swept in bricks of 32^3
    grid points -- a run per brick for one workgroup, the faces towards the finished neighbours gathered into an LDS halo,
    the bricks a coarse wavefront handed out in dependency order (ssor_brick_kernel): whole and partial bricks, a brick
    layer one point thick (33 planes), variable coefficients, all four sweep kinds, the replayed graph.  With missing
    couplings a dependency can lie further back than the brick's ring holds; the handle then keeps its level schedule.
    Same bits as the oracle's sequential sweeps (preconmodule.c:95-193)."""
    from pysparse_amd.device import DeviceSSOR, DeviceSSS
    S = grid_sss(oracle, *grid, seed=11 + sum(grid), keep=keep)
    D = DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)
    K = DeviceSSOR(D, omega, steps)
    nb = -(-grid[0] // 32) * -(-grid[1] // 32) * -(-grid[2] // 32)
    if keep == 1.0:
        assert K.bricks == nb, (K.bricks, nb, K.lds_runs)
    x = rng_vec(S.n, 3)
    y_ref = np.full(S.n, -1.5)
    oracle.ssor_apply(S, x, y_ref, omega, steps)
    y = np.full(S.n, -1.5)
    K.precon(x, y)
    assert np.array_equal(y, y_ref)
    K.precon(x, y)  # the replayed graph
    assert np.array_equal(y, y_ref)


def test_ssor_bricks_refuse_what_is_not_a_grid_operator(oracle):
    """a coupling that wraps around a grid line, a 2-D operator, an operator without the offset -1: no bricks, same bits"""
    from pysparse_amd.device import DeviceSSOR, DeviceSSS
    S = grid_sss(oracle, 128, 128, 64, seed=5, keep=1.0)
    r = 128  # first point of the second grid line: its offset -1 neighbour is the end of the first line
    hi = S.ind[r + 1]
    col = np.insert(S.col, hi, r - 1)
    val = np.insert(S.val, hi, -0.25)
    ind = S.ind.copy()
    ind[r + 1:] += 1
    S2 = oracle.SSS(S.n, val, S.diag, col, ind)
    x = rng_vec(S.n, 3)
    for T in (S2, grid_sss(oracle, 2100, 2100, 0, seed=6, keep=1.0)):
        K = DeviceSSOR(DeviceSSS.from_arrays(T.n, T.ind, T.col, T.val, T.diag), 1.0, 1)
        assert K.bricks == 0
        xx = x[:T.n] if T.n <= x.size else rng_vec(T.n, 4)
        y_ref, y = np.zeros(T.n), np.zeros(T.n)
        oracle.ssor_apply(T, xx, y_ref, 1.0, 1)
        K.precon(xx, y)
        assert np.array_equal(y, y_ref)
