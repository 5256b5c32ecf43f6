import os, sys, time, json
import numpy as np
sys.path.insert(0, '/root/repo')
os.environ["PSP_TUNING"] = "1"
from pysparse_amd import device as dev
for N in (362, 400, 450, 500, 512):
    A = dev.DeviceCSR.poisson(N, N); K = dev.DeviceJacobi(A); n = A.shape[0]; b = np.ones(n)
    row = {}
    for coop in ("1", "0"):
        os.environ["PSP_COOP"] = coop
        for name, solver in (("pcg", dev.pcg), ("minres", dev.minres)):
            x = np.zeros(n); solver(A, b, x, 0.0, 50, K)
            best = 1e9
            for _ in range(3):
                x = np.zeros(n); t = time.perf_counter(); r = solver(A, b, x, 0.0, 2000, K)
                best = min(best, (time.perf_counter() - t) * 1e6 / max(1, min(2000, r[1])))
            row[name + ("_coop" if coop == "1" else "_phase")] = round(best, 2)
    print(N, n, row, flush=True)
