"""world_size 2 and 3 gloo tests (CPU) of the row-partitioned SpMV / PCG driver
(pysparse_amd/distributed.py): partition plan, halo exchange and the two all-reduces per
iteration, with the vector arithmetic supplied by an oracle-backed stand-in."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch = pytest.importorskip("torch")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import oracle as O
        from pysparse_amd import distributed as D
        from dist_oracle_backend import OracleBackend, local_poisson_from_oracle
        be, comm = OracleBackend(), D.Comm()
        out = {}
        if case[0] == "poisson":
            nx, ny, nz = case[1]
            G = O.poisson_csr(nx, ny, nz)
            A = D.DistCSR.poisson(nx, ny, nz, comm, be, local_poisson_from_oracle)
        elif case[0] == "sss":  # an sss_mat on row blocks: rows expanded in sss_matvec's summation order
            rng = np.random.default_rng(5)
            n = 300
            S = O.tendigit_sss(n)
            S.diag[:] = 40.0 + rng.random(n)
            S.val[:] = rng.standard_normal(S.val.size)
            G = S  # the oracle's sss_matvec / sss solvers are the reference here
            A = D.DistCSR.from_global_sss(n, S.ind, S.col, S.val, S.diag, comm, be,
                                          lambda shape, i, c, v: O.CSR(shape, v, c, i))
        else:  # general CSR with irregular coupling across the partition
            rng = np.random.default_rng(5)
            n = 300
            S = O.tendigit_sss(n)  # log-spaced bands: ghosts are scattered index sets
            S.diag[:] = 40.0 + rng.random(n)  # diagonally dominant: rounding is not amplified
            G = O.sss_to_csr(S)
            lo, hi = D.row_range(n, world, rank)
            a, b_ = G.ind[lo], G.ind[hi]
            plan, col_local = D.general_halo_plan(n, lo, hi, G.col[a:b_], world, rank, comm.all_gather_object,
                                                     ind=G.ind[lo:hi + 1] - a)
            A_loc = O.CSR((hi - lo, plan.n_ext), G.val[a:b_], col_local, G.ind[lo:hi + 1] - a)
            A = D.DistCSR(A_loc, plan, comm, be)
        n = G.shape[0]
        lo, hi = A.plan.row_lo, A.plan.row_hi
        # ---- distributed SpMV == rows [lo, hi) of the global SpMV (bit-exact: same row order)
        xg = np.random.default_rng(1).standard_normal(n)
        yg = np.empty(n)
        G.matvec(xg, yg)
        v = A.new_ext()
        A.owned(v).copy_(torch.from_numpy(xg[lo:hi].copy()))
        y = be.zeros(hi - lo)
        A.matvec(v, y)
        out["spmv_ok"] = bool(np.array_equal(y.numpy(), yg[lo:hi]))
        # ---- distributed PCG == oracle PCG on the global system
        bg = np.empty(n)
        G.matvec(np.ones(n), bg)
        dinv_g = O.jacobi_dinv(G.diag if case[0] == "sss" else G.diagonal())
        res = {}
        for name, dg in (("none", None), ("jacobi", dinv_g)):
            xo = np.zeros(n)
            ref = O.pcg(G, bg, xo, 1e-9, 500, dg)
            x = be.zeros(hi - lo)
            hist = []
            got = D.dist_pcg(A, be.from_numpy(bg[lo:hi]), x, 1e-9, 500,
                             be.from_numpy(dg[lo:hi]) if dg is not None else None, hist)
            err = float(np.abs(x.numpy() - xo[lo:hi]).max() / np.abs(xo).max())
            res[name] = (ref, got, err, len(hist))
        out["pcg"] = res
        # lazy x-update loop (default) against the eager one: every truncation point of a run that
        # crosses its stagnation (tol = 0) gives the same (info, iter, relres, x), bit for bit
        bl = be.from_numpy(bg[lo:hi])
        dl = be.from_numpy(dinv_g[lo:hi])
        same, infos = True, set()
        for maxit in list(range(1, 60, 3)) + [400]:
            xe, xl = be.zeros(hi - lo), be.zeros(hi - lo)
            re_ = D._dist_pcg(A, bl, xe, 0.0, maxit, dl)
            rl = D._dist_pcg_lazy(A, bl, xl, 0.0, maxit, dl)
            infos.add(rl[0])
            same = same and tuple(re_) == tuple(rl) and bool(np.array_equal(xe.numpy(), xl.numpy()))
        out["lazy_equals_eager"] = same
        out["lazy_infos"] = sorted(infos)
        # device-resident-scalar driver (here: its Python state mirror, batches of 16 iterations between two
        # reads of the state) against the host-scalar lazy loop at every truncation point, bit for bit --
        # results, iterates and residual histories
        same_dev, dev_infos = True, set()
        for maxit in list(range(1, 60, 3)) + [400]:
            xl, xd = be.zeros(hi - lo), be.zeros(hi - lo)
            hl, hd = [], []
            rl = D._dist_pcg_lazy(A, bl, xl, 0.0, maxit, dl, hl)
            rd = D._dist_pcg_dev(A, bl, xd, 0.0, maxit, dl, hd)
            dev_infos.add(rd[0])
            same_dev = (same_dev and tuple(rl) == tuple(rd) and bool(np.array_equal(xl.numpy(), xd.numpy()))
                        and hl == hd)
        for tol_ in (1e-3, 1e-9):
            xl, xd = be.zeros(hi - lo), be.zeros(hi - lo)
            same_dev = same_dev and tuple(D._dist_pcg_lazy(A, bl, xl, tol_, 500, None)) == \
                tuple(D._dist_pcg_dev(A, bl, xd, tol_, 500, None)) and bool(np.array_equal(xl.numpy(), xd.numpy()))
        out["dev_equals_lazy"] = same_dev
        out["dev_infos"] = sorted(dev_infos)
        # MINRES on row blocks against the oracle's MINRES on the global system
        mres = {}
        for name, dg in (("none", None), ("jacobi", dinv_g)):
            for tol_, mx in ((1e-9, 500), (1e-30, 7)):
                xo = np.zeros(n)
                ref = O.minres(G, bg, xo, tol_, mx, dg, hist=True)
                x = be.zeros(hi - lo)
                hist = []
                got = D.dist_minres(A, be.from_numpy(bg[lo:hi]), x, tol_, mx,
                                    be.from_numpy(dg[lo:hi]) if dg is not None else None, hist)
                err = float(np.abs(x.numpy() - xo[lo:hi]).max() / np.abs(xo).max())
                hr = ref[3][:ref[1] + 1]
                herr = float(np.max(np.abs(np.array(hist) - hr) / hr)) if len(hist) == len(hr) else np.inf
                mres[(name, mx)] = (ref[:3], got, err, herr)
        out["minres"] = mres
        x = be.from_numpy(np.ones(hi - lo))
        out["minres_maxit0"] = D.dist_minres(A, be.from_numpy(bg[lo:hi]), x, 1e-9, 0)
        # maxit exhausted -> iter = maxit + 1; zero rhs -> (0, 0, 0.0)
        x = be.zeros(hi - lo)
        out["maxit"] = D.dist_pcg(A, be.from_numpy(bg[lo:hi]), x, 1e-30, 3)
        x = be.from_numpy(np.ones(hi - lo))
        out["zero"] = D.dist_pcg(A, be.zeros(hi - lo), x, 1e-9, 10) + (float(x.abs().max()),)
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def _worker_breakdown(rank, world, port, case, q):
    """-6 (p.q == 0) and -2 (rho == 0) through the three PCG drivers: same triple, same iterate, same history --
    the slot of the iteration that broke down is never written and must not be reported (no NaN entry)"""
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import oracle as O
        from pysparse_amd import distributed as D
        from dist_oracle_backend import OracleBackend
        be, comm = OracleBackend(), D.Comm()
        n = 64
        out = {}
        for name, diag, dinv_g in (("pq0", np.r_[np.ones(n // 2), -np.ones(n // 2)], None),
                                   ("rho0", np.ones(n), np.r_[np.ones(n // 2), -np.ones(n // 2)]),
                                   # two clean iterations first, then p.q == 0 is not reached: plain convergence
                                   ("spd", np.arange(1.0, n + 1), None)):
            G = O.CSR((n, n), diag, np.arange(n, dtype=np.int32), np.arange(n + 1, dtype=np.int32))
            lo, hi = D.row_range(n, world, rank)
            a, b_ = G.ind[lo], G.ind[hi]
            plan, col_local = D.general_halo_plan(n, lo, hi, G.col[a:b_], world, rank, comm.all_gather_object,
                                                  ind=G.ind[lo:hi + 1] - a)
            A = D.DistCSR(O.CSR((hi - lo, plan.n_ext), G.val[a:b_], col_local, G.ind[lo:hi + 1] - a), plan, comm, be)
            bg = np.ones(n)
            xo = np.zeros(n)
            ho = O.pcg(G, bg, xo, 1e-10, 50, dinv_g, hist=True)
            runs = []
            for fn in (D._dist_pcg, D._dist_pcg_lazy, D._dist_pcg_dev):
                x, h = be.zeros(hi - lo), []
                r = fn(A, be.from_numpy(bg[lo:hi]), x, 1e-10, 50,
                       be.from_numpy(dinv_g[lo:hi]) if dinv_g is not None else None, h)
                runs.append((tuple(r), x.numpy().copy(), h))
            out[name] = {"ref": ho[:3], "runs": [(r, bool(np.array_equal(x, runs[0][1])), h) for r, x, h in runs],
                         "x_ok": bool(np.allclose(runs[0][1], xo[lo:hi], rtol=1e-13, atol=0)),
                         "ref_hist": [float(v) for v in ho[3] if not np.isnan(v)]}
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def _run(world, case, target=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target or _worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    return results


@pytest.mark.parametrize("world,case", [(2, ("poisson", (6, 5, 8))), (3, ("poisson", (5, 4, 7))),
                                        (2, ("poisson", (16, 9, 0))), (2, ("general",)), (3, ("general",)),
                                        (2, ("sss",)), (3, ("sss",))])
def test_row_partitioned_spmv_and_pcg(world, case):
    results = _run(world, case)
    assert len(results) == world
    for rank, out in results.items():
        assert "error" not in out, out.get("error")
        assert out["spmv_ok"]
        for name, (ref, got, err, nhist) in out["pcg"].items():
            assert tuple(got[:2]) == tuple(ref[:2]), (name, ref, got)
            # the recurred ||r|| at the exit iteration sits at the rounding floor of the recurrence
            assert abs(got[2] - ref[2]) <= (1e-6 if case[0] == "poisson" else 0.1) * ref[2]
            assert err < 1e-12
            assert nhist == got[1] + 1
        assert out["lazy_equals_eager"], out["lazy_infos"]
        assert -1 in out["lazy_infos"] and (-5 in out["lazy_infos"] or case[0] != "poisson")
        assert out["dev_equals_lazy"], out["dev_infos"]
        assert -1 in out["dev_infos"] and (-5 in out["dev_infos"] or case[0] != "poisson")
        for key, (ref, got, err, herr) in out["minres"].items():
            assert tuple(got[:2]) == tuple(ref[:2]), (key, ref, got)
            assert abs(got[2] - ref[2]) <= 1e-8 * abs(ref[2]), (key, ref, got)
            assert err < 1e-12 and herr < 1e-8, (key, err, herr)
        assert tuple(out["minres_maxit0"][:2]) == (-1, 0)
        assert tuple(out["maxit"][:2]) == (-1, 4)
        assert out["zero"] == (0, 0, 0.0, 0.0)
    # every rank reports the same triple
    r0 = results[0]["pcg"]["jacobi"][1]
    assert all(results[r]["pcg"]["jacobi"][1] == r0 for r in results)


def test_breakdown_exits_agree_between_the_three_drivers():
    results = _run(2, None, _worker_breakdown)
    for rank, out in results.items():
        assert "error" not in out, out.get("error")
        for name, want in (("pq0", -6), ("rho0", -2), ("spd", 0)):
            o = out[name]
            assert o["ref"][0] == want, (name, o["ref"])
            first = o["runs"][0]
            for r, same_x, h in o["runs"]:
                assert r[:2] == tuple(o["ref"][:2]), (name, r, o["ref"])
                assert r == first[0] and same_x and h == first[2], (name, r, first[0], h, first[2])
                assert not any(v != v for v in h), (name, h)  # no NaN entry for the iteration that broke down
                assert len(h) == len(o["ref_hist"]), (name, h, o["ref_hist"])
            assert o["x_ok"], name


def test_partition_helpers():
    sys.path.insert(0, ROOT)
    from pysparse_amd import distributed as D
    n = 103
    covered = []
    for r in range(8):
        lo, hi = D.row_range(n, 8, r)
        covered.extend(range(lo, hi))
    assert covered == list(range(n))
    # 1024^3 over 8 ranks: 128 planes each, 2^27 rows, halo = one plane on each inner side
    for r in range(8):
        p = D.poisson_halo_plan(1024, 1024, 1024, 8, r)
        assert p.n_owned == 2 ** 27 and p.n_global == 2 ** 30
        assert p.ghost_lo == (0 if r == 0 else 2 ** 20) and p.ghost_hi == (0 if r == 7 else 2 ** 20)
        assert sorted(p.recv) == sorted(p.send) == [q for q in (r - 1, r + 1) if 0 <= q < 8]
    with pytest.raises(ValueError):
        D.poisson_halo_plan(4, 4, 2, 4, 0)


@pytest.mark.parametrize("world", [1, 2, 3, 5])
def test_one_process_driver_partition_equals_the_torch_drivers_plan(world):
    """psp_multi_plan (pysparse_amd/csrc/psp_multi.hip: the partition behind psp_csr_create_multi, pure host code) against
    general_halo_plan of pysparse_amd/distributed.py on the same matrices: same row ranges, same ghost sets, same local column
    numbers, same receive slices and senders' index lists, same ghost-free row range -- so what the gloo tests above establish
    for the one-process-per-GPU driver's partition holds for the one-process driver too (its device side runs under -m gpu)."""
    import ctypes as C
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import oracle as O
    from pysparse_amd import _capi, distributed as D
    import krylov_cases as KC
    L = _capi.lib()
    rng = np.random.default_rng(9)
    S = O.tendigit_sss(700)
    mats = [O.poisson_csr(7, 6, 5), O.sss_to_csr(S), KC.nonsym_csr(O, 333, 4), O.poisson_csr(23, 9)]
    for G in mats:
        n = G.shape[0]
        ranges = [D.row_range(n, world, r) for r in range(world)]
        wanted_all = []
        plans = []
        for r in range(world):  # the torch driver's plan, with its two all-gathers emulated
            lo, hi = ranges[r]
            a, b = G.ind[lo], G.ind[hi]
            calls = []

            def gather(obj, r=r, calls=calls):
                calls.append(obj)
                if len(calls) == 1:
                    return ranges
                return wanted_all if len(wanted_all) == world else [obj if q == r else {} for q in range(world)]
            plan, col_local = D.general_halo_plan(n, lo, hi, G.col[a:b], world, r, gather, ind=G.ind[lo:hi + 1] - a)
            wanted_all.append(calls[1])
            plans.append((plan, col_local))
        for r in range(world):
            plan, col_local = plans[r]
            lo, hi = ranges[r]
            rr = (C.c_int64 * 2)()
            cnt = (C.c_int * 5)()
            ghosts = np.zeros(n, dtype=np.int32)
            links = np.zeros(4 * world, dtype=np.int32)
            lcol = np.zeros(G.ind[hi] - G.ind[lo], dtype=np.int32)
            _capi.check(L.psp_multi_plan(n, n, G.ind.ctypes.data, G.col.ctypes.data, world, r, rr, cnt, ghosts.ctypes.data, n,
                                         links.ctypes.data, world, lcol.ctypes.data))
            assert (rr[0], rr[1]) == (lo, hi)
            assert (cnt[0], cnt[1]) == (plan.ghost_lo, plan.ghost_hi)
            assert (cnt[2], cnt[3]) == tuple(plan.interior) or cnt[3] - cnt[2] == plan.interior[1] - plan.interior[0]
            assert np.array_equal(lcol, col_local)
            ext = np.concatenate([ghosts[:cnt[0]], np.arange(lo, hi), ghosts[cnt[0]:cnt[0] + cnt[1]]])
            assert np.array_equal(ext[col_local], G.col[G.ind[lo]:G.ind[hi]])  # local numbers point at the right entries
            got = {int(links[4 * i]): (int(links[4 * i + 1]), int(links[4 * i + 1] + links[4 * i + 2]), int(links[4 * i + 3]))
                   for i in range(cnt[4])}
            assert set(got) == set(plan.recv)
            for q, (s0, s1) in plan.recv.items():
                assert got[q][:2] == (s0, s1)
                ids = ext[s0:s1] - ranges[q][0]  # what q must send: its owned-local indices
                if got[q][2] >= 0:
                    assert np.array_equal(ids, np.arange(got[q][2], got[q][2] + len(ids)))
                else:
                    assert not np.array_equal(ids, np.arange(ids[0], ids[0] + len(ids)))
