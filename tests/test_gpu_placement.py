"""GPU: placement of the vectors the library owns (pysparse_amd/csrc/psp_place.hip; VERDICT r4 "Next" #3).

The reference's solvers own their work array (pysparse/itsolvers/src/itsolversmodule.c:32-118: `work` is allocated by the
wrapper); here the work vectors that are operands of the iteration's product are drawn among a few candidate buffers and
the best kept (what a bandwidth-bound product takes depends on the pages its operands occupy, profiles/r4_modes.txt).
Other addresses, the same kernels: not a bit of any result may change, the draw happens once per (device, length), small
systems never draw, and psp_set_placement(0) restores plain allocations."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _info(L):
    en, draws, ms = C.c_int(), C.c_longlong(), C.c_double()
    assert L.psp_placement_info(C.byref(en), C.byref(draws), C.byref(ms)) == 0
    return en.value, draws.value, ms.value


def test_solves_with_drawn_work_vectors_keep_their_bits_and_draw_once():
    from pysparse_amd import device as dev
    from pysparse_amd._capi import lib, check
    L = lib()
    grid = (2900, 2900, 0)  # 8.41e6 unknowns: just above the 2^23-element threshold
    A = dev.DeviceCSR.poisson(*grid)
    K = dev.DeviceJacobi(A)
    n = A.shape[0]
    assert n >= 1 << 23
    b = np.random.default_rng(3).standard_normal(n)
    results = {}
    try:
        for on in (0, 1, 1, 0):
            check(L.psp_trim())
            check(L.psp_set_placement(on))
            d0 = _info(L)[1]
            got = []
            for solver in (dev.pcg, dev.minres):
                for Kk in (None, K):
                    x = np.zeros(n)
                    r = solver(A, b, x, 0.0, 12, Kk)
                    got.append((tuple(r[:3]), x))
            drew = _info(L)[1] - d0
            # one draw per length and device: the four solves of a leg share the drawn set through the scratch pool
            assert drew == (1 if on else 0), (on, drew)
            if on in results:
                for (ra, xa), (rb, xb) in zip(results[on], got):
                    assert ra == rb and np.array_equal(xa, xb)
            results[on] = got
        for (ra, xa), (rb, xb) in zip(results[0], results[1]):
            assert ra == rb and np.array_equal(xa, xb)  # placement moves vectors, not bits
    finally:
        check(L.psp_set_placement(1))
        check(L.psp_trim())


def test_small_systems_never_draw_and_the_operand_service():
    from pysparse_amd import device as dev
    from pysparse_amd._capi import lib, check
    L = lib()
    check(L.psp_set_placement(1))
    A = dev.DeviceCSR.poisson(300, 300)
    n = A.shape[0]
    d0 = _info(L)[1]
    x = np.zeros(n)
    dev.pcg(A, np.ones(n), x, 1e-8, 2000, dev.DeviceJacobi(A))
    yp, xp = C.c_void_p(), C.c_void_p()
    rep = (C.c_double * 6)()
    check(L.psp_place_operands(A._h, C.byref(yp), C.byref(xp), rep))
    assert _info(L)[1] == d0 and rep[0] == 0  # plain allocations below the threshold
    out = np.ones(n)
    check(L.psp_memcpy_d2h(out.ctypes.data, yp, 8 * n))
    assert not out.any()  # zeroed
    L.psp_free(yp)
    L.psp_free(xp)
    # at a bandwidth-bound size the service draws and reports what it saw
    B = dev.DeviceCSR.poisson(2900, 2900)
    nb = B.shape[0]
    check(L.psp_place_operands(B._h, C.byref(yp), C.byref(xp), rep))
    assert rep[0] >= 4 and 0 < rep[1] <= rep[2] and 0 < rep[3] <= rep[4] and rep[5] > 0
    xh = np.random.default_rng(0).standard_normal(nb)
    check(L.psp_memcpy_h2d(xp, xh.ctypes.data, 8 * nb))
    B.matvec_dev(xp.value, yp.value)
    y1, y2 = np.empty(nb), np.empty(nb)
    check(L.psp_memcpy_d2h(y1.ctypes.data, yp, 8 * nb))
    B.matvec(xh, y2)  # host-pointer product (its staging pair is drawn too)
    assert np.array_equal(y1, y2)
    L.psp_free(yp)
    L.psp_free(xp)
    check(L.psp_trim())
