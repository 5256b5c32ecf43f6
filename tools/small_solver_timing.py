"""Microseconds per iteration of Jacobi-PCG and Jacobi-MINRES on small 2-D Poisson problems (host-pointer
API, so the PCIe copies of b and x are inside): the launch-latency end of the range.  MINRES with
device-resident scalars (default) against the host-scalar loop (PSP_MINRES_ASYNC=0, run in a child)."""
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    from pysparse_amd import device as dev
    out = {}
    for N in (100, 200, 300, 362, 500, 1000, 2048):
        A = dev.DeviceCSR.poisson(N, N)
        K = dev.DeviceJacobi(A)
        n = A.shape[0]
        b = np.ones(n)
        k = 2000 if N <= 1000 else 400
        row = {}
        for name, solver in (("pcg", dev.pcg), ("minres", dev.minres)):
            x = np.zeros(n)
            solver(A, b, x, 0.0, 50, K)
            best = 1e9
            for _ in range(3):
                x = np.zeros(n)
                t = time.perf_counter()
                r = solver(A, b, x, 0.0, k, K)
                best = min(best, (time.perf_counter() - t) * 1e6 / max(1, min(k, r[1])))
            row[name] = {"us_per_iter": best, "iters": r[1]}
        out["poisson2d(%d)" % N] = row
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print(json.dumps(run()))
    else:
        res = {"device_scalars": run()}
        env = dict(os.environ, PSP_TUNING="1", PSP_COOP="0", PSP_MID="0")  # without the single-kernel loops (psp_coop.hip, psp_mid.hip)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        res["launch_per_phase_loops"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else p.stderr[-500:]
        if isinstance(res["launch_per_phase_loops"], dict):
            for size, row in res["device_scalars"].items():
                a = res["launch_per_phase_loops"][size]
                print("%s: single-kernel loops (default up to 2^20 rows) pcg %.1f / minres %.1f us/it; launch-per-phase loops "
                      "pcg %.1f / minres %.1f" % (size, row["pcg"]["us_per_iter"], row["minres"]["us_per_iter"],
                                                   a["pcg"]["us_per_iter"], a["minres"]["us_per_iter"]), flush=True)
        env = dict(os.environ, PSP_TUNING="1", PSP_COOP="0", PSP_MID="0", PSP_MINRES_ASYNC="0")
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True)
        res["minres_host_scalars"] = json.loads(p.stdout.strip().splitlines()[-1]) if p.returncode == 0 else p.stderr[-500:]
        for size, row in res["device_scalars"].items():
            h = res["minres_host_scalars"][size]["minres"]["us_per_iter"] if isinstance(res["minres_host_scalars"], dict) else float("nan")
            print("%s: pcg %.1f us/it, minres %.1f us/it (host-scalar loop %.1f), minres/pcg %.2f" % (
                size, row["pcg"]["us_per_iter"], row["minres"]["us_per_iter"], h,
                row["minres"]["us_per_iter"] / row["pcg"]["us_per_iter"]), flush=True)
