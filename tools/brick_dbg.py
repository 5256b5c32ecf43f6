import sys, time, faulthandler
faulthandler.dump_traceback_later(100, exit=True)
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import torch  # the test suite's order of loading
from oracle import oracle as O
from pysparse_amd.device import DeviceSSOR, DeviceSSS, DeviceBuffer
from pysparse_amd._capi import lib, check
from test_gpu_ssor import grid_sss
g = tuple(int(t) for t in sys.argv[1].split(","))
omega, steps = float(sys.argv[2]), int(sys.argv[3])
S = grid_sss(O, *g, seed=7 + sum(g), keep=1.0)
D = DeviceSSS.from_arrays(S.n, S.ind, S.col, S.val, S.diag)
K = DeviceSSOR(D, omega, steps); print("bricks", K.bricks, "levels", K.levels, flush=True)
x = np.random.default_rng(3).standard_normal(S.n); yr = np.full(S.n, -1.5)
O.ssor_apply(S, x, yr, omega, steps)
y = np.full(S.n, -1.5)
for rep in range(3):
    K.precon(x, y)
    bad = np.nonzero(y != yr)[0]
    print("apply", rep, "equal", bad.size == 0, "bad rows", bad.size, bad[:6], flush=True)
xd = DeviceBuffer.from_host(x); yd = DeviceBuffer(S.n)
K.precon_dev(xd.ptr, yd.ptr); check(lib().psp_synchronize())
t = time.perf_counter()
for _ in range(5): K.precon_dev(xd.ptr, yd.ptr)
check(lib().psp_synchronize()); print("apply_ms", (time.perf_counter() - t) / 5 * 1e3)
