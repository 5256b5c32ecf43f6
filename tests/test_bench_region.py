"""bench.py's contract-critical part on its own (round 5, VERDICT r4 "Next" #8): the timed region and the numbers of the
JSON line that come out of it.  No GPU: the step, the synchronisation and the events are fakes that record what was done.

  * exactly W untimed and K timed steps; one synchronisation before the clock starts and one before it stops; the two
    events are recorded around the K timed launches, inside the bracket;
  * value = bytes of the whole job per step / wall time per step; roofline.achieved = one GPU's bytes per launch / the
    average launch between the events -- recomputed here from the fakes' numbers;
  * nothing in the region, in the functions it calls, or in the launcher / legs modules can reach the oracle: the oracle
    is imported in bench.py only, inside cpu_baseline / gpu_parity_case / _ref_solve / mtx_leg (the CPU-baseline and
    parity legs), never at module level."""
import ast
import os
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeEvents:
    def __init__(self, log, ms):
        self.log, self._ms = log, ms

    def record(self, i):
        self.log.append(("record", i))

    def ms(self, i, j):
        assert (i, j) == (0, 1)
        return self._ms


@pytest.mark.parametrize("warmup,steps", [(0, 1), (3, 7), (10, 100)])
def test_timed_region_times_exactly_k_steps_between_two_synchronisations(warmup, steps):
    import bench
    log = []

    def step():
        log.append(("step",))
        time.sleep(0.0002)

    def sync():
        log.append(("sync",))
    wall, ev_ms = bench.timed_region(step, sync, FakeEvents(log, 12.5), warmup, steps)
    want = [("step",)] * warmup + [("sync",), ("record", 0)] + [("step",)] * steps + [("record", 1), ("sync",)]
    assert log == want
    assert ev_ms == 12.5
    assert wall >= steps * 0.0002 and wall < steps * 0.0002 + 0.5  # the K timed steps, not the warm-up
    # no device (gloo dry runs): the event time is the wall time
    log.clear()
    wall2, ev2 = bench.timed_region(step, sync, None, warmup, steps)
    assert ev2 == wall2 * 1e3 and ("record", 0) not in log


def test_headline_numbers_are_bytes_over_the_measured_times():
    import bench
    kb_job, kb_launch, wall, ev_ms, steps = 8 * 9932111872, 9932111872, 0.0335, 33.1, 20
    h = bench.headline(kb_job, kb_launch, wall, ev_ms, steps)
    assert h["value"] == kb_job / (wall / steps) / 1e9
    assert h["ms_per_step"] == wall * 1e3 / steps
    assert h["avg_launch_ms"] == ev_ms / steps
    assert h["achieved"] == kb_launch / (ev_ms / steps * 1e-3) / 1e9
    assert h["frac"] == h["achieved"] / 8000.0
    # one GPU: the job's bytes are the launch's bytes, and the two rates differ only by wall vs event time
    h1 = bench.headline(kb_launch, kb_launch, wall, ev_ms, steps)
    assert abs(h1["value"] / h1["achieved"] - (ev_ms * 1e-3) / wall) < 1e-12


def _imports_of(node):
    out = []
    for n in ast.walk(node):
        if isinstance(n, ast.Import):
            out += [a.name for a in n.names]
        elif isinstance(n, ast.ImportFrom):
            out.append(n.module or "")
    return out


def test_the_oracle_is_reachable_from_the_baseline_and_parity_legs_only():
    src = {f: open(os.path.join(ROOT, f)).read() for f in ("bench.py", "bench_common.py", "bench_launch.py", "bench_legs.py")}
    # the launcher, the legs and the shared module never mention it
    for f in ("bench_common.py", "bench_launch.py", "bench_legs.py"):
        assert not any("oracle" in m for m in _imports_of(ast.parse(src[f]))), f
    tree = ast.parse(src["bench.py"])
    allowed = {"cpu_baseline", "mtx_leg"}  # gpu_parity_case / _cpu_case / _ref_solve are handed the module by these
    for node in tree.body:
        imps = [m for m in _imports_of(node) if "oracle" in m]
        if isinstance(node, ast.FunctionDef):
            assert not imps or node.name in allowed, (node.name, imps)
        else:
            assert not imps, "module-level oracle import in bench.py"
    fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    # the region itself: no imports at all, and it calls only what it is handed (+ the clock)
    region = fns["timed_region"]
    assert _imports_of(region) == []
    called = {ast.unparse(c.func) for c in ast.walk(region) if isinstance(c, ast.Call)}
    assert called <= {"step", "sync", "ev.record", "ev.ms", "time.perf_counter", "range"}, called
    assert _imports_of(fns["headline"]) == []
    # the body of a rank reaches the oracle only through cpu_baseline(...) / mtx_leg(...), both after the timed region
    body = fns["run_body"]
    lines = src["bench.py"].splitlines()
    region_line = next(c.lineno for c in ast.walk(body) if isinstance(c, ast.Call) and ast.unparse(c.func) == "timed_region")
    for c in ast.walk(body):
        if isinstance(c, ast.Call) and ast.unparse(c.func) in ("cpu_baseline", "mtx_leg"):
            assert c.lineno > region_line, lines[c.lineno - 1]
