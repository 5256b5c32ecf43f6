#!/usr/bin/env python3
"""tools/h2d_probe.py -- how to get 264 MB of NumPy arrays (an sss_mat at n = 9.3e5) into HBM fastest (round 6, the upload
half of configs[4]'s time-to-solution): plain hipMemcpy from pageable memory, hipHostRegister + copy + unregister, and a
copy through two pinned staging buffers filled by T host threads.  Prints milliseconds for a 158 MB array."""
import ctypes as C
import sys
import threading
import time

import numpy as np

hip = C.CDLL("libamdhip64.so")
vp = C.c_void_p


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> %d" % (what, rc))


def main():
    nbytes = 158691840
    a = np.random.default_rng(0).standard_normal(nbytes // 8)
    src = a.ctypes.data
    d = vp()
    chk(hip.hipMalloc(C.byref(d), C.c_size_t(nbytes)), "hipMalloc")
    chk(hip.hipDeviceSynchronize(), "sync")
    out = {}
    for rep in range(3):
        t0 = time.perf_counter()
        chk(hip.hipMemcpy(d, vp(src), C.c_size_t(nbytes), 1), "memcpy")
        out.setdefault("pageable_ms", []).append((time.perf_counter() - t0) * 1e3)
    for rep in range(3):
        t0 = time.perf_counter()
        chk(hip.hipHostRegister(vp(src), C.c_size_t(nbytes), 0), "register")
        t1 = time.perf_counter()
        chk(hip.hipMemcpy(d, vp(src), C.c_size_t(nbytes), 1), "memcpy")
        t2 = time.perf_counter()
        chk(hip.hipHostUnregister(vp(src)), "unregister")
        t3 = time.perf_counter()
        out.setdefault("register_copy_unregister_ms", []).append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3])
    # staging: two pinned buffers of CH bytes, T threads fill one while the other is on the wire
    for CH, T in ((8 << 20, 4), (16 << 20, 4), (16 << 20, 8), (32 << 20, 8)):
        bufs = []
        for _ in range(2):
            p = vp()
            chk(hip.hipHostMalloc(C.byref(p), C.c_size_t(CH), 0), "hostmalloc")
            bufs.append(p)
        st = vp()
        chk(hip.hipStreamCreateWithFlags(C.byref(st), 1), "stream")
        evs = []
        for _ in range(2):
            e = vp()
            chk(hip.hipEventCreateWithFlags(C.byref(e), 2), "event")
            evs.append(e)

        def fill(dst, off, n):
            per = (n + T - 1) // T
            ths = []
            for t in range(T):
                lo = t * per
                hi = min(n, lo + per)
                if lo >= hi:
                    break
                th = threading.Thread(target=C.memmove, args=(dst + lo, src + off + lo, hi - lo))
                th.start()
                ths.append(th)
            for th in ths:
                th.join()
        for rep in range(3):
            t0 = time.perf_counter()
            k = 0
            for off in range(0, nbytes, CH):
                n = min(CH, nbytes - off)
                b = k & 1
                if k >= 2:
                    chk(hip.hipEventSynchronize(evs[b]), "evsync")
                fill(bufs[b].value, off, n)
                chk(hip.hipMemcpyAsync(vp(d.value + off), bufs[b], C.c_size_t(n), 1, st), "async")
                chk(hip.hipEventRecord(evs[b], st), "rec")
                k += 1
            chk(hip.hipStreamSynchronize(st), "ssync")
            out.setdefault("staged_%dMB_%dthreads_ms" % (CH >> 20, T), []).append((time.perf_counter() - t0) * 1e3)
        for p in bufs:
            hip.hipHostFree(p)
    print(out)


if __name__ == "__main__":
    main()
