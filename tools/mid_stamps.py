import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
from pysparse_amd import device as dev
for grid in ((724,724,0),(1024,1024,0)):
    A = dev.DeviceCSR.poisson(*grid); n = A.shape[0]
    b = np.random.default_rng(1).standard_normal(n); x = np.zeros(n)
    for wgsel in ("0", "100"):
        os.environ["PSP_MID_STAMPS"] = wgsel
        print(grid, "wg", wgsel, dev.pcg(A, b, x, 0.0, 20, dev.DeviceJacobi(A))[:2], flush=True)
