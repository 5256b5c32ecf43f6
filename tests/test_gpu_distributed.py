"""GPU: the row-partitioned driver with the real HIP backend.

(1) world 1, in-process: dist_pcg(HipBackend) == psp_pcg on the same operator.
(2) two ranks = two processes sharing cuda:0.  The product transport is RCCL (backend
    "nccl"), which needs one GPU per rank; on the one-GPU test box the ranks talk through a
    test-only Comm that stages tensors through host memory over gloo.  Everything else --
    slab generator, halo plan, extended-vector layout, kernels, reduction protocol -- is
    the product code path."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch = pytest.importorskip("torch")


def test_world1_matches_single_gpu_solver(oracle):
    from pysparse_amd import device as dev, distributed as D
    be = D.HipBackend(0)
    try:
        comm = D.SingleComm()
        nx, ny, nz = 40, 36, 28
        A = D.DistCSR.poisson(nx, ny, nz, comm, be, dev.DeviceCSR.poisson_slab)
        n = A.n_local
        G = oracle.poisson_csr(nx, ny, nz)
        b_np = np.empty(n)
        G.matvec(np.ones(n), b_np)
        dinv_np = oracle.jacobi_dinv(G.diagonal())
        for dinv in (None, dinv_np):
            xo = np.zeros(n)
            ref = oracle.pcg(G, b_np, xo, 1e-9, 1000, dinv)
            x = be.zeros(n)
            got = D.dist_pcg(A, be.from_numpy(b_np), x, 1e-9, 1000, be.from_numpy(dinv) if dinv is not None else None)
            assert tuple(got[:2]) == tuple(ref[:2])
            assert np.abs(x.cpu().numpy() - xo).max() <= 1e-12 * np.abs(xo).max()
            # the three drivers -- device-resident scalars (default), host-scalar lazy, host-scalar eager --
            # are the same algorithm: bitwise identical results, iterates and histories at every truncation
            bt = be.from_numpy(b_np)
            dt = be.from_numpy(dinv) if dinv is not None else None
            for tol, mx in [(1e-9, 1000)] + [(0.0, k) for k in (1, 2, 3, 15, 16, 17, 33, 50)]:
                res = []
                for fn in (D._dist_pcg_dev, D._dist_pcg_lazy, D._dist_pcg):
                    xx, hh = be.zeros(n), []
                    r = fn(A, bt, xx, tol, mx, dt, hh)
                    res.append((tuple(r), xx.cpu().numpy().copy(), hh))
                for other in res[1:]:
                    assert other[0] == res[0][0] and np.array_equal(other[1], res[0][1]) and other[2] == res[0][2]
            # MINRES on row blocks == the oracle's MINRES
            for tol, mx in ((1e-9, 1000), (1e-30, 9)):
                xo = np.zeros(n)
                refm = oracle.minres(G, b_np, xo, tol, mx, dinv, hist=True)
                x = be.zeros(n)
                hh = []
                gotm = D.dist_minres(A, bt, x, tol, mx, dt, hh)
                assert tuple(gotm[:2]) == tuple(refm[:2])
                assert abs(gotm[2] - refm[2]) <= 1e-8 * refm[2]
                assert np.abs(x.cpu().numpy() - xo).max() <= 1e-12 * np.abs(xo).max()
                hr = refm[3][:refm[1] + 1]
                assert len(hh) == len(hr) and np.max(np.abs(np.array(hh) - hr) / hr) <= 1e-8
    finally:
        from pysparse_amd import _capi
        _capi.lib().psp_set_stream(None)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, transport="host_staged"):
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from oracle import oracle as O
        from pysparse_amd import device as dev, distributed as D

        class HostStagedComm(D.Comm):
            def allreduce_sum(self, t):
                h = t.cpu()
                dist.all_reduce(h)
                t.copy_(h)
                return t

            def exchange(self, sends, recvs):
                torch.cuda.synchronize()
                hs = [(p, t.cpu()) for p, t in sends]
                hr = [(p, torch.empty(t.shape, dtype=t.dtype)) for p, t in recvs]
                ops = [dist.P2POp(dist.irecv, t, p) for p, t in hr] + [dist.P2POp(dist.isend, t, p) for p, t in hs]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                for (p, t), (_, h) in zip(recvs, hr):
                    t.copy_(h)

            def exchange_start(self, sends, recvs):
                # start the host-staged transfer now, finish it inside wait(): the interior
                # SpMV launch sits between the two, like with the RCCL transport
                torch.cuda.synchronize()
                hs = [(p, t.cpu()) for p, t in sends]
                hr = [(p, torch.empty(t.shape, dtype=t.dtype)) for p, t in recvs]
                ops = [dist.P2POp(dist.irecv, t, p) for p, t in hr] + [dist.P2POp(dist.isend, t, p) for p, t in hs]
                works = dist.batch_isend_irecv(ops) if ops else []

                def wait():
                    for w in works:
                        w.wait()
                    for (p, t), (_, h) in zip(recvs, hr):
                        t.copy_(h)
                return wait

        be = D.HipBackend(0)
        # "gloo_device": the product's Comm class unchanged -- gloo moves device tensors on this image, so the
        # batch_isend_irecv / all_reduce calls are the ones the RCCL runs make
        comm = HostStagedComm() if transport == "host_staged" else D.Comm()
        nx, ny, nz = 48, 40, 24
        A = D.DistCSR.poisson(nx, ny, nz, comm, be, dev.DeviceCSR.poisson_slab)
        G = O.poisson_csr(nx, ny, nz)
        n = G.shape[0]
        lo, hi = A.plan.row_lo, A.plan.row_hi
        out = {}
        xg = np.random.default_rng(2).standard_normal(n)
        yg = np.empty(n)
        G.matvec(xg, yg)
        v = A.new_ext()
        A.owned(v).copy_(be.from_numpy(xg[lo:hi]))
        y = be.zeros(hi - lo)
        A.matvec(v, y)
        out["spmv_ok"] = bool(np.array_equal(y.cpu().numpy(), yg[lo:hi]))
        bg = np.empty(n)
        G.matvec(np.ones(n), bg)
        dinv_g = O.jacobi_dinv(G.diagonal())
        xo = np.zeros(n)
        ref = O.pcg(G, bg, xo, 1e-9, 1000, dinv_g)
        x = be.zeros(hi - lo)
        got = D.dist_pcg(A, be.from_numpy(bg[lo:hi]), x, 1e-9, 1000, be.from_numpy(dinv_g[lo:hi]))
        out["pcg"] = (ref, got, float(np.abs(x.cpu().numpy() - xo[lo:hi]).max() / np.abs(xo).max()))
        xl = be.zeros(hi - lo)
        gl = D._dist_pcg_lazy(A, be.from_numpy(bg[lo:hi]), xl, 1e-9, 1000, be.from_numpy(dinv_g[lo:hi]))
        out["dev_equals_lazy"] = bool(tuple(gl) == tuple(got) and torch.equal(xl, x))
        xo = np.zeros(n)
        refm = O.minres(G, bg, xo, 1e-9, 1000, dinv_g)
        xm = be.zeros(hi - lo)
        gotm = D.dist_minres(A, be.from_numpy(bg[lo:hi]), xm, 1e-9, 1000, be.from_numpy(dinv_g[lo:hi]))
        out["minres"] = (refm, gotm, float(np.abs(xm.cpu().numpy() - xo[lo:hi]).max() / np.abs(xo).max()))
        # the operator of the strong-scaling runs (index-free slab, psp_csr_poisson_big_slab) on the same ranks:
        # same bits for the product, same PCG result
        B = D.DistCSR.poisson(nx, ny, nz, comm, be, dev.DeviceCSR.poisson_big_slab)
        yb = be.zeros(hi - lo)
        vb = B.new_ext()
        B.owned(vb).copy_(be.from_numpy(xg[lo:hi]))
        B.matvec(vb, yb)
        xb = be.zeros(hi - lo)
        gb = D.dist_pcg(B, be.from_numpy(bg[lo:hi]), xb, 1e-9, 1000, be.from_numpy(dinv_g[lo:hi]))
        out["big_slab"] = bool(torch.equal(yb, y) and tuple(gb) == tuple(got) and torch.equal(xb, x))
        out["shake_injected"] = int(getattr(D, "SHAKE_INJECTED", 0))  # > 0 only under PSP_DIST_SHAKE (test_gpu_shake.py)
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


@pytest.mark.parametrize("transport", ["host_staged", "gloo_device"])
@pytest.mark.parametrize("world", [2, 3])
def test_ranks_sharing_one_gpu(world, transport):
    """world 3: the middle rank has ghost planes on both sides (the interior ranks of the 8-GPU partition)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, transport)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    for rank, out in results.items():
        assert "error" not in out, out.get("error")
        assert out["spmv_ok"]
        ref, got, err = out["pcg"]
        assert tuple(got[:2]) == tuple(ref[:2])
        assert err < 1e-12
        assert out["dev_equals_lazy"] and out["big_slab"]
        ref, got, err = out["minres"]
        assert tuple(got[:2]) == tuple(ref[:2]) and err < 1e-12
