"""CPU: the threading model of libpysparse_hip.so (pysparse_amd/csrc/psp_internal.h, "Threading model"; SURVEY 8b "one HIP
stream per handle; handles not thread-safe"): the context an entry point enqueues on belongs to the calling thread, and an
entry point locks the HANDLES it is given -- in address order, recursively -- so two threads that share a handle take turns
and two threads with different handles do not wait for each other.  No GPU needed: psp_thread_info and the lock hook
psp_debug_hold_handles run without one."""
import ctypes as C
import threading
import time

from pysparse_amd import _capi


def info(L):
    slot, dev, s = C.c_int(-1), C.c_int(-1), C.c_void_p()
    assert L.psp_thread_info(C.byref(slot), C.byref(dev), C.byref(s)) == 0
    return slot.value, dev.value, s.value


def test_every_thread_has_its_own_slot_and_slots_are_reused():
    L = _capi.lib()
    main = info(L)
    assert main[0] >= 0 and info(L) == main  # stable within a thread
    seen, barrier = [], threading.Barrier(4)

    def worker():
        a = info(L)
        barrier.wait()  # all four alive at once: the slots must differ
        seen.append(a[0])
        assert info(L)[0] == a[0]

    ts = [threading.Thread(target=worker) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(set(seen)) == 4 and main[0] not in seen and all(s > 0 for s in seen)
    later = []
    t = threading.Thread(target=lambda: later.append(info(L)[0]))
    t.start()
    t.join()
    assert later[0] in seen  # a slot given back by a thread that ended is handed out again (its workspaces stay cached)


def _hold(L, a, b, ms, out, k):
    t = time.perf_counter()
    assert L.psp_debug_hold_handles(a, b, ms) == 0
    out[k] = time.perf_counter() - t


def test_threads_that_share_a_handle_take_turns_and_others_overlap():
    L = _capi.lib()
    h = [C.c_void_p(0x1000 * (i + 1)) for i in range(4)]  # the lock table is keyed by address; nothing is dereferenced
    out = {}
    # the same handle from two threads: the second waits for the first
    t0 = time.perf_counter()
    ts = [threading.Thread(target=_hold, args=(L, h[0], None, 150, out, k)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert time.perf_counter() - t0 >= 0.29
    # different handles: side by side
    t0 = time.perf_counter()
    ts = [threading.Thread(target=_hold, args=(L, h[k], None, 150, out, k)) for k in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert time.perf_counter() - t0 < 0.29
    # an operator shared by two solves with different preconditioners: they take turns on the operator
    t0 = time.perf_counter()
    ts = [threading.Thread(target=_hold, args=(L, h[0], h[1 + k], 150, out, k)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert time.perf_counter() - t0 >= 0.29


def test_opposite_lock_orders_do_not_deadlock():
    """thread 1 is handed (a, b), thread 2 (b, a), many times over: address order inside HandleLock means neither can hold one
    and wait for the other; the same handle twice in one call is locked once (recursive)"""
    L = _capi.lib()
    a, b = C.c_void_p(0x7000), C.c_void_p(0x3000)
    done = []

    def run(x, y):
        for _ in range(300):
            assert L.psp_debug_hold_handles(x, y, 0) == 0
        assert L.psp_debug_hold_handles(x, x, 0) == 0
        done.append(1)

    ts = [threading.Thread(target=run, args=(a, b)), threading.Thread(target=run, args=(b, a)),
          threading.Thread(target=run, args=(b, a)), threading.Thread(target=run, args=(a, b))]
    [t.start() for t in ts]
    [t.join(timeout=60) for t in ts]
    assert len(done) == 4
