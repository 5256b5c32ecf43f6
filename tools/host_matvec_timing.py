#!/usr/bin/env python3
"""Host-pointer A.matvec(x, y) (the reference boundary, csr_mat.c:141-163: NumPy buffers in, NumPy buffer out) against
what the PCIe link of this box gives: per call 8 n bytes up + 8 n bytes down + one kernel.
Prints JSON: ms per call and GB/s for the pipelined (default) and the plain path (PSP_HOST_PIPELINE=0 under
PSP_TUNING=1, child process), next to the link's ceilings measured in the same job (pinned copies, torch)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(grid):
    from pysparse_amd import device as dev
    A = dev.DeviceCSR.poisson(*grid)
    n = A.shape[0]
    x = np.random.default_rng(0).standard_normal(n)
    y = np.empty(n)
    A.matvec(x, y)
    A.matvec(x, y)
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        A.matvec(x, y)
        ts.append(time.perf_counter() - t)
    xd, yd = dev.DeviceBuffer.from_host(x), dev.DeviceBuffer(n)
    A.matvec_dev(xd.ptr, yd.ptr)
    same = bool(np.array_equal(y, yd.download()))
    print(json.dumps({"n": n, "ms": min(ts) * 1e3, "ms_all": [round(t * 1e3, 2) for t in ts], "bit_equal_to_device_product": same,
                      "kernel": A.kernel_info()[0]}))


def link_ceilings():
    code = ("import torch,time,json\nn=(1<<30)//8\nd=torch.empty(n,dtype=torch.float64,device='cuda')\ne=torch.empty_like(d)\n"
            "p=torch.empty(n,dtype=torch.float64).pin_memory()\nq=torch.empty(n,dtype=torch.float64).pin_memory()\n"
            "s1,s2=torch.cuda.Stream(),torch.cuda.Stream()\n"
            "def t(f):\n f();torch.cuda.synchronize();b=9e9\n for _ in range(3):\n  a=time.perf_counter();f();torch.cuda.synchronize();b=min(b,time.perf_counter()-a)\n return b\n"
            "def both():\n with torch.cuda.stream(s1): d.copy_(p,non_blocking=True)\n with torch.cuda.stream(s2): q.copy_(e,non_blocking=True)\n"
            "G=(1<<30)/1e9\nprint(json.dumps({'pinned_h2d_GBps':G/t(lambda:d.copy_(p,non_blocking=True)),'pinned_d2h_GBps':G/t(lambda:q.copy_(e,non_blocking=True)),'pinned_duplex_sum_GBps':2*G/t(both)}))")
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
        return json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(tuple(int(t) for t in sys.argv[2].split(",")))
        sys.exit(0)
    res = {"link": link_ceilings(), "cases": {}}
    for name, grid in (("512^3", "512,512,512"), ("4096^2", "4096,4096,0")):
        for mode, env in (("pipelined", {}), ("plain", {"PSP_TUNING": "1", "PSP_HOST_PIPELINE": "0"})):
            e = dict(os.environ)
            e.update(env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", grid], env=e, capture_output=True,
                               text=True, timeout=600)
            try:
                r = json.loads(p.stdout.strip().splitlines()[-1])
            except Exception:  # noqa: BLE001
                r = {"error": (p.stdout + p.stderr)[-400:]}
            if "ms" in r:
                gb = 8.0 * r["n"] / 1e9
                r["GBps_each_way_if_sequential"] = 2 * gb / (r["ms"] * 1e-3) / 2  # bytes one way / half the time
                r["GBps_total_both_ways"] = 2 * gb / (r["ms"] * 1e-3)
                lk = res["link"]
                if "pinned_duplex_sum_GBps" in lk:
                    r["frac_of_duplex_ceiling"] = r["GBps_total_both_ways"] / lk["pinned_duplex_sum_GBps"]
                    seq = 1.0 / (1.0 / lk["pinned_h2d_GBps"] + 1.0 / lk["pinned_d2h_GBps"]) * 2
                    r["frac_of_sequential_pinned_ceiling"] = r["GBps_total_both_ways"] / seq
            res["cases"]["%s %s" % (name, mode)] = r
    print(json.dumps(res, indent=1))
