"""GPU parity at BASELINE.json's FULL sizes (C2 4096^2, C3 512^3, one rank's slab of C4 1024^3),
through properties that need no CPU oracle at that size:

  * exact row sums: A*ones is 0 in the interior and the number of missing neighbours on the
    boundary (small integers, so the comparison is exact);
  * every SpMV kernel (w4 / w3 / w2, and the sss_mat product from the lower triangle only) gives
    the SAME BITS for a random x -- the small-size tests pin each of them to the oracle bit for bit;
  * symmetry (A x).z == x.(A z) and linearity to rounding;
  * Jacobi-PCG: the recurred residual norm the solver reports equals ||b - A x|| recomputed from
    the returned x (pcg.c:146-153 keeps the recurred one), identical iteration counts and iterates
    <= 1e-12 apart whichever kernel multiplies; the 2-D config run to convergence recovers x = 1.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W3 = (1 << 20) + 16578
W2 = 16578


def missing_neighbours(nx, ny, nz):
    shape = (nz, ny, nx) if nz else (ny, nx)
    e = np.zeros(shape)
    for ax in range(len(shape)):
        sl_lo = [slice(None)] * len(shape)
        sl_hi = [slice(None)] * len(shape)
        sl_lo[ax], sl_hi[ax] = 0, -1
        e[tuple(sl_lo)] += 1
        e[tuple(sl_hi)] += 1
    return e.ravel()


def dev_dot(L, check, n, a, b, out):
    check(L.psp_k_dot(n, a.ptr, b.ptr, out.ptr))
    return float(out.download()[0])


@pytest.mark.parametrize("grid", [(4096, 4096, 0), (512, 512, 512)])
def test_fullsize_spmv_properties(grid):
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()
    nx, ny, nz = grid
    A = dev.DeviceCSR.poisson(nx, ny, nz)
    n = A.shape[0]
    stencil = 7 if nz else 5
    faces = 2 * (nx * ny + ny * nz + nx * nz) if nz else 2 * (nx + ny)
    assert n == nx * ny * max(nz, 1) and A.nnz == stencil * n - faces  # test_spmatrix.py:77-78 extended to 3-D
    assert A.kernel_info()[0] == "csr_spmv_w4"

    ones = dev.DeviceBuffer.from_host(np.ones(n))
    y = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, y.ptr)
    assert np.array_equal(y.download(), missing_neighbours(nx, ny, nz))
    del ones

    rng = np.random.default_rng(5)
    xh = rng.standard_normal(n)
    x = dev.DeviceBuffer.from_host(xh)
    A.matvec_dev(x.ptr, y.ptr)
    y_w4 = y.download()
    for variant, name in ((W3, "csr_spmv_w3"), (W2, "csr_spmv_w2")):
        A.set_variant(variant)
        assert A.kernel_info()[0] == name
        y.zero()
        A.matvec_dev(x.ptr, y.ptr)
        assert np.array_equal(y.download(), y_w4), name
    A.set_variant(-1)

    # spot rows against the definition (corners, edges, interior), exact
    nxy = nx * ny
    for k in sorted({k for k in (0, 1, nx - 1, nx, nxy - 1, nxy, n // 2 + nx // 3, n - nxy - 1, n - nx, n - 1) if 0 <= k < n}):
        i, j, l = k % nx, (k // nx) % ny, k // nxy
        s = 0.0
        if nz and l > 0:
            s += -1.0 * xh[k - nxy]
        if j > 0:
            s += -1.0 * xh[k - nx]
        if i > 0:
            s += -1.0 * xh[k - 1]
        s += (6.0 if nz else 4.0) * xh[k]
        if i < nx - 1:
            s += -1.0 * xh[k + 1]
        if j < ny - 1:
            s += -1.0 * xh[k + nx]
        if nz and l < nz - 1:
            s += -1.0 * xh[k + nxy]
        assert y_w4[k] == s, k

    # sss_mat: same operator from the strict lower triangle; same per-row order => same bits
    S = dev.DeviceSSS.poisson(nx, ny, nz)
    assert S.kernel_info()[0] == "sss_spmv_w4"
    y.zero()
    S.matvec_dev(x.ptr, y.ptr)
    assert np.array_equal(y.download(), y_w4)
    del S

    # symmetry and linearity, to rounding
    z = dev.DeviceBuffer.from_host(rng.standard_normal(n))
    az = dev.DeviceBuffer(n)
    A.matvec_dev(z.ptr, az.ptr)
    A.matvec_dev(x.ptr, y.ptr)
    out = dev.DeviceBuffer(4)
    ax_z = dev_dot(L, check, n, y, z, out)
    x_az = dev_dot(L, check, n, x, az, out)
    scale = np.sqrt(dev_dot(L, check, n, y, y, out) * dev_dot(L, check, n, z, z, out))
    assert abs(ax_z - x_az) <= 1e-12 * scale
    azh = az.download()
    comb = dev.DeviceBuffer.from_host(2.5 * xh - 0.75 * z.download())
    A.matvec_dev(comb.ptr, az.ptr)
    lin = az.download() - (2.5 * y_w4 - 0.75 * azh)
    assert np.abs(lin).max() <= 64 * np.finfo(float).eps * (np.abs(y_w4).max() + np.abs(azh).max())


def _pcg_dev(L, check, dev, A, K, n, x, b, tol, maxit):
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    check(L.psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, tol, maxit, C.byref(info), C.byref(it), C.byref(rr), None))
    check(L.psp_synchronize())
    return info.value, it.value, rr.value


def test_fullsize_pcg_512_fixed_iterations_consistent_across_kernels():
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()
    nx = ny = nz = 512
    A = dev.DeviceCSR.poisson(nx, ny, nz)
    n = A.shape[0]
    K = dev.DeviceJacobi(A)
    ones = dev.DeviceBuffer.from_host(np.ones(n))
    b = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, b.ptr)  # b = A*ones (demo_pcg.py:57-58)
    del ones
    out = dev.DeviceBuffer(4)
    n2b = np.sqrt(dev_dot(L, check, n, b, b, out))
    r = dev.DeviceBuffer(n)
    results = {}
    for name, variant in (("w4", -1), ("w3", W3)):
        A.set_variant(variant)
        x = dev.DeviceBuffer(n)
        x.zero()
        info, it, relres = _pcg_dev(L, check, dev, A, K, n, x, b, 0.0, 25)
        assert (info, it) == (-1, 26)  # tol = 0: the loop runs out, iter = maxit + 1 (pcg.c:165)
        # true residual of the returned iterate against the recurred norm the solver kept
        A.matvec_dev(x.ptr, r.ptr)
        check(L.psp_k_residual(n, b.ptr, r.ptr, None, out.ptr))
        true_rel = np.sqrt(float(out.download()[0])) / n2b
        assert abs(true_rel - relres) <= 1e-10 * relres
        results[name] = (relres, x.download())
        del x
    A.set_variant(-1)
    S = dev.DeviceSSS.poisson(nx, ny, nz)
    KS = dev.DeviceJacobi(S)
    x = dev.DeviceBuffer(n)
    x.zero()
    info, it, relres = _pcg_dev(L, check, dev, S, KS, n, x, b, 0.0, 25)
    assert (info, it) == (-1, 26)
    results["sss"] = (relres, x.download())
    ref_rr, ref_x = results["w4"]
    assert 0 < ref_rr < 0.2  # 25 Jacobi-PCG iterations take the residual well below ||b||
    for name in ("w3", "sss"):
        rr, xs = results[name]
        # same products in the same per-row order: only the dot-product partial order differs
        assert abs(rr - ref_rr) <= 1e-12 * ref_rr, name
        assert np.abs(xs - ref_x).max() <= 1e-12 * np.abs(ref_x).max(), name


def test_fullsize_pcg_4096sq_converges_to_ones():
    """C2: 2-D 5-pt 4096^2, b = A*ones, Jacobi-PCG to 1e-8 (the demo_pcg.py flow at full size)."""
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()
    A = dev.DeviceCSR.poisson(4096, 4096)
    n = A.shape[0]
    K = dev.DeviceJacobi(A)
    ones = dev.DeviceBuffer.from_host(np.ones(n))
    b = dev.DeviceBuffer(n)
    A.matvec_dev(ones.ptr, b.ptr)
    x = dev.DeviceBuffer(n)
    x.zero()
    info, it, relres = _pcg_dev(L, check, dev, A, K, n, x, b, 1e-8, 40000)
    assert info == 0 and relres <= 1e-8
    assert 2000 < it < 30000  # O(N) iterations for the N x N grid (G1: 160 at N = 100, tol 1e-6)
    xs = x.download()
    assert np.abs(xs - 1.0).max() < 0.1  # error <= cond * relres ~ 7e6 * 1e-8, far from tight
    # the recurred residual is the true one
    r = dev.DeviceBuffer(n)
    out = dev.DeviceBuffer(4)
    n2b = np.sqrt(dev_dot(L, check, n, b, b, out))
    A.matvec_dev(x.ptr, r.ptr)
    check(L.psp_k_residual(n, b.ptr, r.ptr, None, out.ptr))  # r := b - A x, out[0] = r.r
    true_rel = np.sqrt(float(out.download()[0])) / n2b
    assert abs(true_rel - relres) <= 1e-4 * relres  # thousands of recurrence steps apart


def test_fullsize_slab_of_1024_cubed():
    """C4: the row block one of 8 ranks owns (128 z-planes of 1024^3 = 2^27 rows, ghost-extended
    columns): interior rank => every row sum is 0 except nothing (all six neighbours exist inside
    the global grid in z; x/y faces are boundary)."""
    from pysparse_amd import device as dev
    nx = ny = 1024
    nz = 1024
    nxy = nx * ny
    lo, hi = 3 * 128 * nxy, 4 * 128 * nxy  # rank 3 of 8
    shift = lo - nxy
    ncols = (hi - lo) + 2 * nxy
    A = dev.DeviceCSR.poisson_slab(nx, ny, nz, lo, hi, shift, ncols)
    assert A.shape == (1 << 27, ncols) and A.nnz == 938999808  # interior ranks
    assert A.kernel_info()[0] == "csr_spmv_w4"
    ones = dev.DeviceBuffer.from_host(np.ones(ncols))
    y = dev.DeviceBuffer(hi - lo)
    A.matvec_dev(ones.ptr, y.ptr)
    e2 = missing_neighbours(nx, ny, 0)  # x/y faces only: the z neighbours are ghosts that exist
    yh = y.download().reshape(128, nxy)
    assert np.array_equal(yh, np.broadcast_to(e2, (128, nxy)))
    xh = np.random.default_rng(2).standard_normal(ncols)
    x = dev.DeviceBuffer.from_host(xh)
    A.matvec_dev(x.ptr, y.ptr)
    y4 = y.download()
    # the multi-GPU product: interior rows first, the two ghost-touching plane ranges after the halo
    # wait -- same bits, and the fused dot equals the separate one to rounding
    from pysparse_amd._capi import check, lib
    L = lib()
    out = dev.DeviceBuffer(4)
    for variant in (-1, W3):
        A.set_variant(variant)
        y.zero()
        from pysparse_amd._capi import WAIT_FN
        check(L.psp_k_csr_matvec_overlap(A._h, x.ptr, nxy, y.ptr, nxy, (hi - lo) - nxy, WAIT_FN(0), None, out.ptr))
        assert np.array_equal(y.download(), y4), variant
        fused = float(out.download()[0])
        sep = float(np.dot(xh[nxy:nxy + (hi - lo)], y4))
        assert abs(fused - sep) <= 1e-10 * abs(sep)
    A.set_variant(W3)
    assert A.kernel_info()[0] == "csr_spmv_w3"
    y.zero()
    A.matvec_dev(x.ptr, y.ptr)
    assert np.array_equal(y.download(), y4)
    k = 77 * nxy + 513 * nx + 100  # an interior row, against the definition
    g = k + nxy  # its position in the extended vector
    s = 0.0
    for c, v in ((g - nxy, -1.0), (g - nx, -1.0), (g - 1, -1.0), (g, 6.0), (g + 1, -1.0), (g + nx, -1.0), (g + nxy, -1.0)):
        s += v * xh[c]
    assert y4[k] == s


@pytest.mark.parametrize("grid", [(40, 30, 20), (64, 50, 0), (129, 7, 3)])
def test_poisson_big_equals_csr_operator(oracle, grid):
    """psp_csr_poisson_big (w4 layout only, no CSR arrays) against the ordinary generator: same bits for
    y = A x, y = A^T x, the diagonal and a Jacobi-PCG solve; download is refused"""
    from pysparse_amd import device as dev
    from pysparse_amd._capi import PspError
    A = dev.DeviceCSR.poisson(*grid)
    B = dev.DeviceCSR.poisson_big(*grid)
    assert B.shape == A.shape and B.nnz == A.nnz and B.kernel_info()[0] == "csr_spmv_w4"
    n = A.shape[0]
    x = np.random.default_rng(1).standard_normal(n)
    ya, yb = np.empty(n), np.empty(n)
    A.matvec(x, ya)
    B.matvec(x, yb)
    assert np.array_equal(ya, yb)
    assert np.array_equal(A.diagonal(), B.diagonal())
    b = np.empty(n)
    A.matvec(np.ones(n), b)
    xa, xb = np.zeros(n), np.zeros(n)
    ra = dev.pcg(A, b, xa, 1e-10, 2000, dev.DeviceJacobi(A))
    rb = dev.pcg(B, b, xb, 1e-10, 2000, dev.DeviceJacobi(B))
    # the same operator in two layouts: identical counts; bit-identical iterates when both run the same loop (an operator
    # with CSR arrays and <= 2^18 rows takes the single-kernel loop of psp_coop.hip, the index-free one never does)
    assert ra[:2] == rb[:2] and abs(ra[2] - rb[2]) <= 1e-6 * rb[2]
    assert np.abs(xa - xb).max() <= 1e-12 * np.abs(xb).max()
    if n > (1 << 18):
        assert ra == rb and np.array_equal(xa, xb)
    with pytest.raises(PspError):
        B.download()
    ta, tb, to = np.empty(n), np.empty(n), np.empty(n)
    A.matvec_transp(x, ta)  # both through csr_spmv_w4_transp: exact, no atomics
    B.matvec_transp(x, tb)
    oracle.poisson_csr(*grid).matvec_transp(x, to)
    assert np.array_equal(ta, to) and np.array_equal(tb, to)
    B.set_variant(16578)  # asking for a CSR kernel changes nothing: there are no CSR arrays
    B.matvec(x, yb)
    assert np.array_equal(ya, yb)


def test_fullsize_1024_cubed_on_one_gpu():
    """C4's operator on ONE GPU (n = 2^30, nnz = 7.5e9 > 2^31: index-free layout only): exact row sums,
    spot rows against the definition, 5 Jacobi-PCG iterations with a consistent residual"""
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()
    N = 1024
    A = dev.DeviceCSR.poisson_big(N, N, N)
    n = A.shape[0]
    assert n == 1 << 30 and A.nnz == 7 * n - 6 * N * N == 7509901312
    ones = dev.DeviceBuffer(n)
    y = dev.DeviceBuffer(n)
    chunk = np.ones(1 << 26)  # ones on the device without an 8 GiB host array
    for k in range(0, n, chunk.size):
        check(L.psp_memcpy_h2d(ones.ptr + 8 * k, chunk.ctypes.data, 8 * chunk.size))
    A.matvec_dev(ones.ptr, y.ptr)
    # row sums plane by plane: interior planes see only the x/y faces, the first and last one more
    e2 = missing_neighbours(N, N, 0)
    plane = np.empty(N * N)
    for l in (0, 1, 511, 1022, 1023):
        check(L.psp_memcpy_d2h(plane.ctypes.data, y.ptr + 8 * l * N * N, 8 * N * N))
        assert np.array_equal(plane, e2 + (1.0 if l in (0, N - 1) else 0.0)), l
    b = y  # b = A*ones
    x = dev.DeviceBuffer(n)
    x.zero()
    K = dev.DeviceJacobi(A)
    aop, kop = dev._Op(A, "matvec"), dev._Op(K, "precon")
    info, it, rr = C.c_int(), C.c_int(), C.c_double()
    check(L.psp_pcg_dev(aop._h, kop._h, n, x.ptr, b.ptr, 0.0, 5, C.byref(info), C.byref(it), C.byref(rr), None))
    check(L.psp_synchronize())
    assert (info.value, it.value) == (-1, 6) and 0 < rr.value < 1
    r = ones  # reuse
    out = dev.DeviceBuffer(4)
    A.matvec_dev(x.ptr, r.ptr)
    check(L.psp_k_residual(n, b.ptr, r.ptr, None, out.ptr))
    rr_true = np.sqrt(float(out.download()[0]))
    check(L.psp_k_dot(n, b.ptr, b.ptr, out.ptr))
    n2b = np.sqrt(float(out.download()[0]))
    assert abs(rr_true / n2b - rr.value) <= 1e-10 * rr.value


@pytest.mark.parametrize("grid,parts", [((24, 18, 12), 3), ((40, 30, 0), 4), ((33, 5, 9), 2)])
def test_poisson_big_slab_equals_csr_slab(oracle, grid, parts):
    """psp_csr_poisson_big_slab (index-free row slab in extended-vector coordinates: what a rank of the
    strong-scaling runs holds) against psp_csr_poisson_slab and the oracle's global rows: same bits for
    the product (whole and split around a halo wait), the fused dot, and the diagonal"""
    from pysparse_amd import device as dev, distributed as D
    from pysparse_amd._capi import check, lib
    L = lib()
    nx, ny, nz = grid
    G = oracle.poisson_csr(nx, ny, nz)
    n = G.shape[0]
    xg = np.random.default_rng(4).standard_normal(n)
    yg = np.empty(n)
    G.matvec(xg, yg)
    for rank in range(parts):
        plan = D.poisson_halo_plan(nx, ny, nz, parts, rank)
        shift = plan.row_lo - plan.ghost_lo
        A = dev.DeviceCSR.poisson_slab(nx, ny, nz, plan.row_lo, plan.row_hi, shift, plan.n_ext)
        B = dev.DeviceCSR.poisson_big_slab(nx, ny, nz, plan.row_lo, plan.row_hi, shift, plan.n_ext)
        assert B.shape == A.shape and B.nnz == A.nnz and B.kernel_info()[0] == "csr_spmv_w4"
        xe = xg[shift:shift + plan.n_ext].copy()
        ya, yb = np.empty(plan.n_owned), np.empty(plan.n_owned)
        A.matvec(xe, ya)
        B.matvec(xe, yb)
        assert np.array_equal(ya, yg[plan.row_lo:plan.row_hi]) and np.array_equal(yb, ya)
        # split product + fused dot (psp_k_csr_matvec_overlap), as the multi-GPU driver calls it
        xd = dev.DeviceBuffer.from_host(xe)
        yd = dev.DeviceBuffer(plan.n_owned)
        out = dev.DeviceBuffer(2)
        from pysparse_amd._capi import WAIT_FN
        cb = WAIT_FN(lambda ctx: 0)
        dots = []
        for M in (A, B):
            yd.zero()
            check(L.psp_k_csr_matvec_overlap(M._h, xd.ptr, plan.p_offset, yd.ptr, plan.interior[0], plan.interior[1],
                                             cb, None, out.ptr))
            assert np.array_equal(yd.download(), ya)
            dots.append(float(out.download()[0]))
        assert dots[0] == dots[1]
        assert abs(dots[0] - float(np.dot(xg[plan.row_lo:plan.row_hi], ya))) <= 1e-12 * abs(dots[0])


def test_poisson_big_slab_rejects_bad_arguments():
    from pysparse_amd import device as dev
    from pysparse_amd._capi import PspError
    with pytest.raises(PspError):
        dev.DeviceCSR.poisson_big_slab(8, 8, 8, 64, 128, 64, 64 + 8)  # halo below the slab not covered
    with pytest.raises(PspError):
        dev.DeviceCSR.poisson_big_slab(8, 8, 8, 128, 64, 0, 512)       # empty / inverted range


def _release_this_process_gpu_memory():
    """handles of earlier tests that await garbage collection, the solvers' scratch pool (up to a third of the device) and
    torch's caching allocator all belong to THIS process; a test that starts children at configs[3]'s true size frees them"""
    import gc
    gc.collect()
    from pysparse_amd._capi import lib
    lib().psp_trim()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
    except ImportError:
        pass


def test_config4_true_rank_shares_torch_ranks_rehearsal():
    """BASELINE.json configs[3] on a four-rank run's true shares: 1024 x 1024 x 512 cut into 2 z-slabs of 2^28 rows (1.9e9
    nonzeros each: beyond 32-bit CSR offsets, index-free slab operator), bench.py's own launcher and driver, both ranks on
    this one GPU over gloo (RCCL needs a GPU per rank).  20 Jacobi-PCG iterations of the row-partitioned driver must
    reproduce the residual of the whole problem solved by the single-GPU loop (`strong_n1`, timed by rank 0 in the same job)
    to rounding.  (Rounds 2-5 ran four such ranks = the whole 1024^3 here, 71 s; the one-process test below still does, and
    VERDICT r5 #5 asked for no more one-GPU rehearsals: half the problem, the same per-rank path.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    _release_this_process_gpu_memory()  # the children need ~200 of the 288 GB: this process must not sit on its caches
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--share-gpu", "--grid", "1024,1024,512", "--steps", "3", "--warmup", "1", "--pcg-iters", "20",
                          "--no-cpu-baseline", "--no-clocks"], capture_output=True, text=True, cwd=root, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    sys.path.insert(0, root)
    import bench_line
    line, d = bench_line.read(out.stdout)  # the printed line (<= 6 KB) and the full record in the side file it names
    assert line["roofline"]["pcg_iters_per_s"] > 0 and 1.9 < line["predicted"]["vs_n1"] < 2.0 and line["vs_n1"] > 0
    assert d["config"]["n"] == 1 << 29 and d["config"]["nnz"] == 7 * (1 << 29) - 2 * (1 << 20) - 4 * (1 << 19)
    assert d["config"]["rows_per_gpu"] == 1 << 28
    assert d["launcher"]["stage"] == "torch_rccl_ranks", d["launcher"]  # (what the torch ranks died of, if they did)
    assert d["rccl_ranks"] == 2 and d["scaling"] == "strong"
    assert d["parity_vs_n1"]["ok"], d["parity_vs_n1"]
    one, two = d["strong_n1"]["pcg_check"], d["pcg_check"]
    assert (one["info"], one["iter"]) == (two["info"], two["iter"]) == (-1, 21)
    assert abs(one["relres"] - two["relres"]) <= 1e-12 * one["relres"]


def test_config4_1024_cubed_one_process_device_list_rehearsal():
    """The same configs[3] operator through the ONE-process driver behind the C ABI (psp_csr_poisson_multi, psp_multi.hip):
    four ranks = four entries of the device list, all device 0 here (2^28 rows, 1.9e9 nonzeros per rank), bench.py
    --single-process.  20 Jacobi-PCG iterations of the four ranks must reproduce the one-GPU solve of the same 1024^3 system
    that the same job runs first (`strong_n1`: relres and two checksums of x within 1e-9, equal (info, iter): the line's
    own `parity_vs_n1`), and the line must judge itself against the prediction (`predicted`).  (Rounds 3-5 ran the problem a
    second time on ONE rank of the same driver for the comparison: 11 s more for what `strong_n1` already gives.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    _release_this_process_gpu_memory()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--single-process", "--share-gpu",
                          "--steps", "3", "--warmup", "1", "--pcg-iters", "16"], capture_output=True, text=True,
                         cwd=root, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    sys.path.insert(0, root)
    import bench_line
    line, four = bench_line.read(out.stdout)
    assert four["config"]["n"] == 1 << 30 and four["config"]["nnz"] == 7509901312
    assert four["ranks"] == 4 and four["config"]["rows_per_gpu"] == 1 << 28 and "dry_run" in four
    assert (four["pcg_check"]["info"], four["pcg_check"]["iter"]) == (-1, 21)
    par = four["parity_vs_n1"]
    assert par["ok"] and par["same_info_iter"] and par["max_rel_diff"] <= 1e-9, par
    one = four["strong_n1"]
    assert one["grid"] == [1024, 1024, 1024] and one["pcg_check"]["info"] == -1
    assert 3.8 < line["predicted"]["vs_n1"] < 4.0 and "missed_budget" in line["predicted"] and line["vs_n1"] > 0


@pytest.mark.gpu
def test_host_pointer_matvec_pipeline_is_the_device_product():
    """A.matvec(x, y) on NumPy buffers of a large offset-structured operator runs chunked -- x going up, row blocks
    launched as their x entries arrive, y coming down, all at once (psp_csr.hip host_matvec_pipelined; the reference
    boundary csr_mat.c:141-163).  Same kernel per row: the bits of the whole-vector device product; the chunk edges
    (rows whose offsets reach into the next chunk, the last partial chunk) are where a mistake would show."""
    from pysparse_amd import device as dev
    for grid in ((512, 512, 270), (8192, 8200, 0)):  # 70.8e6 rows (17 chunks, last partial), 67.2e6 rows in 2-D
        A = dev.DeviceCSR.poisson(*grid)
        n = A.shape[0]
        assert n >= (1 << 26) and A.kernel_info()[0] == "csr_spmv_w4"
        x = np.random.default_rng(3).standard_normal(n)
        y = np.full(n, np.nan)
        A.matvec(x, y)
        xd, yd = dev.DeviceBuffer.from_host(x), dev.DeviceBuffer(n)
        A.matvec_dev(xd.ptr, yd.ptr)
        assert np.array_equal(y, yd.download())
        y2 = np.full(n, np.nan)
        A.matvec(x, y2)  # staging vectors are reused between calls
        assert np.array_equal(y, y2)
        A.matvec(np.ones(n), y)  # exact row sums: the number of missing neighbours (small integers)
        assert np.array_equal(y, missing_neighbours(*grid).ravel())


@pytest.mark.gpu
def test_sss_from_arrays_at_c3_size(oracle):
    """sss_from_arrays at 512^3 (1.3e8 rows, 4.0e8 lower entries handed over as host arrays): the mirror is built with one
    wave per row, which at this size used to ask for a launch of 8.6e9 threads ("invalid configuration argument"; the
    generators never go that way).  Same bits as the device-generated operator for a random x; precon.ssor on it (bricks)
    applies."""
    from pysparse_amd import device as dev
    nx = 512
    S = oracle.poisson_sss(nx, nx, nx)
    D = dev.DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)
    G = dev.DeviceSSS.poisson(nx, nx, nx)
    n = S.n
    x = dev.DeviceBuffer.from_host(np.random.default_rng(2).standard_normal(n))
    y1, y2 = dev.DeviceBuffer(n), dev.DeviceBuffer(n)
    D.matvec_dev(x.ptr, y1.ptr)
    G.matvec_dev(x.ptr, y2.ptr)
    assert np.array_equal(y1.download(), y2.download())
    del G, y2
    K = dev.DeviceSSOR(D, 1.0, 1)
    assert K.bricks == 16 ** 3
    K.precon_dev(x.ptr, y1.ptr)
    z = y1.download()
    assert np.isfinite(z).all() and abs(z).max() > 0


@pytest.mark.gpu
def test_dropin_module_holds_1024_cubed_on_one_gpu():
    """`spmatrix.poisson_csr(1024, 1024, 1024)` without a device list: 7.5e9 stored entries do not fit the 32-bit CSR of
    csr_mat.h:6-13, so the module builds the index-free operator with 64-bit row offsets (psp_csr_poisson_big) -- the one-GPU
    baseline of configs[3] -- and `nnz` reads through psp_csr_nnz64.  Exact row sums, a few Jacobi-PCG / MINRES iterations
    through krylov with host vectors."""
    from pysparse.itsolvers import krylov
    from pysparse.precon import precon
    from pysparse.sparse import spmatrix
    N = 1024
    A = spmatrix.poisson_csr(N, N, N)
    n = N ** 3
    assert A.shape == (n, n) and A.nnz == 7 * n - 6 * N * N
    e = np.ones(n)
    b = np.empty(n)
    A.matvec(e, b)
    assert b[0] == 3.0 and b[n - 1] == 3.0 and b[n // 2 + N * N // 2 + N // 2] == 0.0
    assert b.sum() == 6.0 * N * N  # one missing neighbour per boundary face point
    K = precon.jacobi(A, 1.0, 1)
    x = np.zeros(n)
    info, it, rr = krylov.pcg(A, b, x, 1e-30, 4, K)
    assert (info, it) == (-1, 5) and 0.0 < rr < 1.0 and np.isfinite(x).all()
    x[:] = 0.0
    info, it, rr = krylov.minres(A, b, x, 1e-30, 4, K)
    assert (info, it) == (-1, 4) and 0.0 < rr < 1.0 and np.isfinite(x).all()
