"""GPU against the reference's iterates AT BASELINE.json's OWN SIZES (configs[1] 4096^2, configs[2] 512^3).

tests/test_gpu_fullsize.py checks the full-size runs through properties and cross-kernel comparisons; the golden-size
tests pin every kernel to the oracle bit for bit / to 1e-12.  This file closes the gap in between: the same k
iterations (tol = 0, so exactly k run) of Jacobi-PCG (pysparse/itsolvers/src/pcg.c:91-163) and Jacobi-MINRES
(minres.c:96-193) on the SAME system (b = A*ones, x0 = 0)

  * by the oracle (oracle/pysparse_oracle.c, one core),
  * by the compiled reference examples/poisson_test/pcg.c when oracle/_ref was built (it travels to the GPU box),
  * on the GPU through the host-pointer entry points the drop-in modules call,

and requires equal (info, iter) and relres / max-norm of x within bench.parity_bound(n, k) = 32 k sqrt(n) eps (eps = 2^-52) -- the
size-dependent form of north_star's 1e-12: two CPU implementations that differ only in the order of their dot-product
sums (the oracle's sequential loops, the reference with OpenBLAS) are themselves 4.9e-11 apart at 512^3 after 3
iterations (BENCH_r03.json, `iterates_max_rel_diff`), where the bound gives 2.5e-10.  The same comparison rides in
every default bench.py run as `parity_check`."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1e6
    return 0.0


@pytest.mark.parametrize("grid,k", [((4096, 4096, 0), 10), ((512, 512, 512), 3)])
def test_gpu_iterates_against_oracle_and_reference_at_baseline_sizes(oracle, grid, k):
    import bench
    from pysparse_amd import device as dev
    if grid[2] and _mem_available_gb() < 48:
        pytest.skip("512^3 on the host (14 GB of matrix + vectors) needs more memory than this box has free")
    out = bench.gpu_parity_case(dev, oracle, grid, k)
    n = out["n"]
    assert n == grid[0] * grid[1] * max(grid[2], 1)
    bound = bench.parity_bound(n, k)
    assert bound < 5e-10  # the bar stays tight: 2.9e-10 at C2 / k = 10, 2.5e-10 at C3 / k = 3
    for name in ("pcg", "minres"):
        r = out[name]
        # tol = 0: PCG runs out with iter = maxit + 1 (pcg.c:165), MINRES stops at it_max (minres.c:114)
        assert r["info_iter_gpu"] == r["info_iter_oracle"] == ([-1, k + 1] if name == "pcg" else [-1, k]), (name, r)
        assert r["relres_rel_diff"] <= bound, (name, r)
        assert r["x_max_rel_diff"] <= bound, (name, r)
    if oracle.have_ref():
        v = out["pcg"]["vs_reference_pcg"]
        assert v["info_iter_reference"] == [-1, k + 1]
        assert v["x_max_rel_diff"] <= bound and v["relres_rel_diff"] <= bound, v
        assert v["oracle_vs_reference_x_max_rel_diff"] <= bound  # the CPU-vs-CPU floor lies inside the same bound
    assert out["ok"]
