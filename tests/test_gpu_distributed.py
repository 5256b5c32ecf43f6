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
    finally:
        from pysparse_amd import _capi
        _capi.lib().psp_set_stream(None)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
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
        comm = HostStagedComm()
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
        q.put((rank, out))
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, {"error": traceback.format_exc()}))


def test_two_ranks_sharing_one_gpu():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    for rank, out in results.items():
        assert "error" not in out, out.get("error")
        assert out["spmv_ok"]
        ref, got, err = out["pcg"]
        assert tuple(got[:2]) == tuple(ref[:2])
        assert err < 1e-12
