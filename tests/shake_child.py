"""Child process of tests/test_gpu_shake.py (started with PSP_TUNING=1): drives the one-process multi-device driver
(psp_multi.hip) under delay injection and prints ONE JSON line.

  race    a fixed 3 ms delay in front of rank 1's vector clearing, (a) with the round-4 fix reverted (the copy stream's
          wait for the receiver's own stream): the product must come back WRONG -- the stress finds the race it is there
          to find -- and (b) on HEAD: the product is the oracle's
  stress  random delays of 0..max_us at every cut point: products, PCG and MINRES on 2..5 ranks sharing the GPU must be
          bit-identical to the same call without delays, `runs` times per configuration

Reference loops being sharded: pcg.c:91-163, minres.c:96-193; product csr_mat.c:49-54."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402
from pysparse_amd import _capi, device as dev  # noqa: E402

L = _capi.lib()
ALL = 0xFFFFFFFF
P_CLEAR = 1 << 8  # psp_internal.h kShakeClear


def arm(seed, lo, hi, points=ALL, ranks=ALL, revert=0):
    _capi.check(L.psp_debug_shake(seed, lo, hi, points, ranks, revert))


def disarm():
    _capi.check(L.psp_debug_shake(-1, 0, 0, 0, 0, 0))


def injected():
    n = C.c_longlong(0)
    _capi.check(L.psp_debug_shake_count(C.byref(n)))
    return int(n.value)


def race():
    grid = (12, 11, 10)
    G = O.poisson_csr(*grid)
    n = G.shape[0]
    x = np.random.default_rng(0).standard_normal(n)
    yo = np.empty(n)
    G.matvec(x, yo)
    AM = dev.DeviceCSR.poisson_multi(*grid, devices=[0, 0, 0])
    out = {}
    for name, revert in (("reverted", 1), ("head", 0)):
        wrong = 0
        for k in range(5):
            arm(k, 3000, 3000, P_CLEAR, 1 << 1, revert)
            y = np.empty(n)
            AM.matvec(x, y)
            wrong += 0 if np.array_equal(y, yo) else 1
        out[name + "_wrong_of_5"] = wrong
        out[name + "_injected"] = injected()
    disarm()
    y = np.empty(n)
    AM.matvec(x, y)
    out["disarmed_ok"] = bool(np.array_equal(y, yo))
    print(json.dumps(out))


def stress(runs, max_us):
    out = {"runs_per_config": runs, "max_us": max_us, "configs": 0, "mismatches": [], "injected": 0, "shaken_calls": 0}
    seed = 0
    for ranks, grid in ((2, (16, 9, 0)), (3, (12, 11, 10)), (5, (6, 5, 8)), (4, (24, 20, 0)), (3, (20, 6, 7))):
        G = O.poisson_csr(*grid)
        n = G.shape[0]
        AM = dev.DeviceCSR.poisson_multi(*grid, devices=[0] * ranks)
        K = dev.DeviceJacobi(AM)
        x = np.random.default_rng(1).standard_normal(n)
        b = np.empty(n)
        G.matvec(np.ones(n), b)

        def calls():
            y = np.empty(n)
            AM.matvec(x, y)
            res = [("matvec", (), y)]
            for solver in (dev.pcg, dev.minres):
                for Kk in (None, K):
                    for tol, maxit in ((0.0, 13), (1e-10, 400)):
                        xs = np.zeros(n)
                        r = solver(AM, b, xs, tol, maxit, Kk)
                        res.append((solver.__name__, tuple(r[:3]), xs))
            return res

        disarm()
        base = calls()
        yo = np.empty(n)
        G.matvec(x, yo)
        assert np.array_equal(base[0][2], yo)
        for k in range(runs):
            arm(seed, 0, max_us)
            seed += 1
            got = calls()
            out["shaken_calls"] += len(got)
            for (nm, ra, va), (_, rb, vb) in zip(base, got):
                if ra != rb or not np.array_equal(va, vb):
                    out["mismatches"].append({"ranks": ranks, "grid": grid, "call": nm, "seed": seed - 1, "base": ra, "got": rb})
            out["injected"] += injected()
            if k % 10 == 9:  # a long campaign must show that it lives (the box ends a silent command)
                print("[shake] ranks %d grid %s: %d / %d runs, %d delays injected, %d mismatches" % (
                    ranks, grid, k + 1, runs, out["injected"], len(out["mismatches"])), file=sys.stderr, flush=True)
        out["configs"] += 1
    disarm()
    out["mismatches"] = out["mismatches"][:10]
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "race":
        race()
    else:
        stress(int(sys.argv[2]), int(sys.argv[3]))
