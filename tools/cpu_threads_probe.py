#!/usr/bin/env python3
"""How many host threads does bench.py's labelled "all host cores" line get, and what do they give?  Prints the affinity
mask, the cgroup CPU quota and the row-parallel oracle SpMV (orc_csr_matvec_threads) at C2 for 1 ... 128 threads.
GPU box, round 4: affinity 256, cpu.max 1600000/100000 = 16 cores -> 31.7 GB/s (1) ... 216 GB/s (16), flat beyond."""
import sys,time,os,numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import oracle as O
print("usable", bench._usable_cores(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max","/sys/fs/cgroup/cpu/cpu.cfs_quota_us","/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except OSError as e: print(f, "absent")
A=O.poisson_csr(4096,4096,0)
n=A.shape[0]
x=np.random.default_rng(0).standard_normal(n);y=np.empty(n)
for t in (1,4,8,16,32,64,128):
    A.matvec_threads(x,y,t)
    ts=[]
    for _ in range(5):
        t0=time.perf_counter(); A.matvec_threads(x,y,t); ts.append(time.perf_counter()-t0)
    print(t, "%.2f ms %.1f GB/s"%(np.median(ts)*1e3, bench.csr_model_bytes(n,A.nnz)/np.median(ts)/1e9), flush=True)
