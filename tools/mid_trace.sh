#!/bin/bash
# rocprofv3 --kernel-trace --stats of single-kernel solves (psp_mid.hip): 1000 iterations of Jacobi-PCG / Jacobi-MINRES at
# 1024^2, 724^2, 512^2 (tol = 0) -> gpurun_out/r5_mid_kernel_stats.csv; the kernels' durations / 1000 are the per-iteration
# times tools/mid_ab.py measures from the host
set -u
OUT=gpurun_out/mid_trace; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > $OUT/run.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from pysparse_amd import device as dev
for N in (1024, 724, 512):
    A = dev.DeviceCSR.poisson(N, N); K = dev.DeviceJacobi(A); n = A.shape[0]
    b = np.random.default_rng(1).standard_normal(n)
    for s in (dev.pcg, dev.minres):
        x = np.zeros(n); print(N, s.__name__, s(A, b, x, 0.0, 1000, K)[:2], flush=True)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $OUT/run.py > $OUT/run.log 2>&1
cp $OUT/t/*/*kernel_stats.csv gpurun_out/r5_mid_kernel_stats.csv
cat $OUT/run.log | tail -6
grep -i "mid_kernel" gpurun_out/r5_mid_kernel_stats.csv | cut -c1-60,300-
python3 - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r5_mid_kernel_stats.csv")):
    if "mid_kernel" in r["Name"]:
        nm = r["Name"].split("(")[0].replace("void (anonymous namespace)::", "").replace("void psp::(anonymous namespace)::", "")
        print("%-40s calls %s  total %.3f ms  per call %.3f ms" % (nm, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
rm -rf $OUT
