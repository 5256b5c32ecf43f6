"""GPU: randomized structure stress of y = A x -- every applicable kernel (default selection, w3, w2,
stream) must give the oracle's bits on hundreds of small random matrices: offset-structured (w4 / w4x
eligible) with random offset sets, banded (w3 eligible), scattered, tiny and rectangular shapes, empty
rows / columns, rows longer than a tile.  Seeds are fixed: failures are reproducible."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W3, W2, STREAM = (1 << 20) + 16578, 16578, 0
W3_VARIANTS = (W3, W3 + (1 << 25), W3 + (2 << 25), W3 + (3 << 25))


def make_matrix(O, rng):
    kind = rng.choice(["offsets", "offsets", "banded", "scattered", "tiny", "longrow"])
    if kind == "tiny":
        m, n = int(rng.integers(1, 6)), int(rng.integers(1, 6))
    elif kind == "longrow":
        m, n = int(rng.integers(2, 40)), int(rng.integers(3000, 9000))
    else:
        m = int(rng.integers(1, 3000))
        n = m if rng.random() < 0.5 else int(rng.integers(1, 3000))
    rows = []
    if kind == "offsets":
        k = int(rng.integers(1, 33))
        span = int(rng.choice([3, 40, 1000, 5000]))
        offs = np.unique(rng.integers(-span, span + 1, size=k))
        keep = rng.choice([0.5, 0.9, 1.0])
        for r in range(m):
            c = r + offs
            c = c[(c >= 0) & (c < n)]
            rows.append(c[rng.random(c.size) < keep] if rng.random() > 0.03 else c[:0])
    elif kind == "banded":
        hb = int(rng.integers(1, 200))
        mr = int(rng.integers(1, 30))
        for r in range(m):
            ctr = min(r * n // max(m, 1), n - 1)
            lo, hi = max(0, ctr - hb), min(n, ctr + hb + 1)
            L = int(rng.integers(0, min(mr, hi - lo) + 1))
            rows.append(np.sort(rng.choice(np.arange(lo, hi), size=L, replace=False)))
    elif kind == "longrow":
        for r in range(m):
            L = int(rng.choice([0, 5, 600, 2500, n]))
            rows.append(np.sort(rng.choice(n, size=min(L, n), replace=False)))
    else:
        mr = int(rng.integers(1, 12))
        for r in range(m):
            L = int(rng.integers(0, min(mr, n) + 1))
            rows.append(np.sort(rng.choice(n, size=L, replace=False)))
    ind = np.zeros(m + 1, dtype=np.int32)
    np.cumsum([len(c) for c in rows], out=ind[1:])
    col = (np.concatenate(rows) if ind[-1] else np.zeros(0)).astype(np.int32)
    val = rng.standard_normal(ind[-1])
    return kind, O.CSR((m, n), val, col, ind)


@pytest.mark.parametrize("block", range(8))
def test_random_structures_all_kernels_bit_exact(oracle, block):
    from pysparse_amd.device import DeviceCSR
    rng = np.random.default_rng(1000 + block)
    seen = {}
    for t in range(40):
        kind, A = make_matrix(oracle, rng)
        m, n = A.shape
        D = DeviceCSR.from_arrays(A.shape, A.ind, A.col, A.val)
        x = rng.standard_normal(n)
        y_ref = np.full(m, 7.0)
        A.matvec(x, y_ref)
        for variant in (-1, W2, STREAM, 16594, 210, 147, 129, 165) + W3_VARIANTS:  # + waves-per-workgroup / layout knobs
            D.set_variant(variant)
            name = D.kernel_info()[0]
            y = np.full(m, -3.0)
            D.matvec(x, y)
            assert np.array_equal(y, y_ref), (block, t, kind, A.shape, A.nnz, variant, name)
            if variant == -1:
                seen[name] = seen.get(name, 0) + 1
        xt = rng.standard_normal(m)
        yt_ref = np.full(n, 9.0)
        A.matvec_transp(xt, yt_ref)
        for variant in (-1, W2):
            D.set_variant(variant)
            yt = np.full(n, -9.0)
            D.matvec_transp(xt, yt)
            assert np.array_equal(yt, yt_ref), (block, t, kind, A.shape, variant)
    assert len(seen) >= 2, seen  # the selection really visits several kernels


@pytest.mark.parametrize("block", range(4))
def test_random_symmetric_structures_sss_bit_exact(oracle, block):
    from pysparse_amd.device import DeviceSSS
    rng = np.random.default_rng(2000 + block)
    for t in range(30):
        n = int(rng.integers(2, 2500))
        structured = rng.random() < 0.6
        rows = []
        if structured:
            k = int(rng.integers(1, 9))
            offs = -np.unique(rng.integers(1, int(rng.choice([3, 60, 1500])) + 1, size=k))
            keep = rng.choice([0.7, 1.0])
            for r in range(n):
                c = np.sort(r + offs)
                c = c[c >= 0]
                rows.append(c[rng.random(c.size) < keep])
        else:
            for r in range(n):
                L = int(rng.integers(0, min(8, r) + 1))
                rows.append(np.sort(rng.choice(r, size=L, replace=False)) if L else np.zeros(0, dtype=np.int64))
        ind = np.zeros(n + 1, dtype=np.int32)
        np.cumsum([len(c) for c in rows], out=ind[1:])
        col = (np.concatenate(rows) if ind[-1] else np.zeros(0)).astype(np.int32)
        S = oracle.SSS(n, rng.standard_normal(ind[-1]), rng.standard_normal(n), col, ind)
        D = DeviceSSS.from_arrays(n, S.ind, S.col, S.val, S.diag)
        x = rng.standard_normal(n)
        y_ref = np.full(n, 1.5)
        S.matvec(x, y_ref)
        for variant in (-1, W3, W2):
            D.set_variant(variant)
            y = np.full(n, -2.5)
            D.matvec(x, y)
            assert np.array_equal(y, y_ref), (block, t, n, structured, variant, D.kernel_info())


def test_no_device_memory_leak_across_handle_lifecycles(oracle):
    """create / use / destroy every kind of handle (csr with each kernel's side tables, sss, jacobi,
    ssor, the solvers' scratch) repeatedly: free device memory returns to where it started"""
    import ctypes as C
    import gc
    from pysparse_amd import device as dev
    from pysparse_amd._capi import check, lib
    L = lib()

    def free_bytes():
        gc.collect()
        check(L.psp_synchronize())
        check(L.psp_trim())
        f, t = C.c_int64(), C.c_int64()
        check(L.psp_mem_info(C.byref(f), C.byref(t)))
        return f.value

    def cycle():
        A = dev.DeviceCSR.poisson(96, 96, 96)
        n = A.shape[0]
        x, y = np.ones(n), np.empty(n)
        for v in (-1, (1 << 20) + 16578, 16578, 0):  # w4, w3, w2, stream tables
            A.set_variant(v)
            A.matvec(x, y)
        A.set_variant((1 << 20) + 16578)
        A.set_schedule(512)
        A.matvec(x, y)
        A.set_variant(-1)
        S = dev.DeviceSSS.poisson(96, 96, 96)
        S.matvec(x, y)
        b = y.copy()
        xs = np.zeros(n)
        dev.pcg(A, b, xs, 1e-6, 200, dev.DeviceJacobi(A))
        dev.pcg(S, b, xs, 1e-6, 50, dev.DeviceSSOR(S))
        dev.minres(S, b, xs, 1e-6, 200, dev.DeviceJacobi(S))
        B = dev.DeviceCSR.poisson_big(64, 64, 64)
        B.matvec(np.ones(B.shape[0]), np.empty(B.shape[0]))
        R = oracle.poisson_csr(20, 20)
        dev.DeviceCSR.from_arrays(R.shape, R.ind, R.col, R.val).matvec_transp(np.ones(400), np.empty(400))

    cycle()  # first use builds process-lifetime things (workspace, pinned scalars)
    f0 = free_bytes()
    for _ in range(3):
        cycle()
    f1 = free_bytes()
    assert abs(f1 - f0) <= 8 << 20, (f0, f1)  # allocator granularity, not a leak of any table (tens of MB each)
