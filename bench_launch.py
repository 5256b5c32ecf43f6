"""bench_launch.py -- how an N > 1 run of bench.py gets its ranks and survives their failures (round 4; split out of
bench.py in round 5 so that the contract-critical part -- the timed region and the line -- can be audited on its own):

  * `python bench.py --gpus N` as a plain script: `orchestrate` runs a LADDER of fresh child processes, each in its own
    process group with a time-out, inside --deadline; the first stage that prints a valid line wins, else ONE error line;
  * `python -m torch.distributed.run ... bench.py --gpus N` (the driver's way): every rank guards itself (`RankGuard`).

Nothing here touches the GPU or measures anything."""
import json
import os
import socket
import subprocess
import sys
import time

from bench_common import METRIC, ROOT  # noqa: F401
from bench_line import attach_launcher, compact_launcher

# The ladder of an N > 1 run started as a plain script: every stage is a FRESH child process (this process never
# touches the GPU, and a process that has is never re-executed); the first stage that prints a valid line wins.
#   torch_rccl_ranks     one torch.distributed rank per GPU; halos = RCCL send/recv, reductions = RCCL all-reduce
#   single_process_rccl  ONE process, device list (psp_csr_poisson_multi); halos = peer copies, reductions = RCCL
#                        inside the library (ncclCommInitAll)
#   single_process_fold  the same with the reductions through the fold kernel over peer pointers (no RCCL at all)
LADDER = ("torch_rccl_ranks", "single_process_rccl", "single_process_fold")


STAGE_CAP_S = {"torch_rccl_ranks": 300.0, "single_process_rccl": 200.0, "single_process_fold": 200.0}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _stage_cmd(stage, argv, n):
    """(command, extra environment) of one ladder stage"""
    me = os.path.join(ROOT, "bench.py")  # the stages are bench.py runs
    if stage == "torch_rccl_ranks":
        return ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
                 "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), me] + argv + ["--stage", stage], {})
    env = {}
    if stage == "single_process_fold":
        env = {"PSP_TUNING": "1", "PSP_MULTI_REDUCE": "local"}
    return [sys.executable, me] + argv + ["--single-process", "--stage", stage], env


def _run_stage(cmd, env, timeout_s, log):
    """run one stage in its own process group; (rc, stdout, stderr tail, wall seconds, timed_out).  On a time-out
    the whole group is ended -- SIGTERM, then SIGKILL -- by its group id: the ranks are grandchildren."""
    import signal
    import tempfile
    t0 = time.time()
    with tempfile.TemporaryFile() as fo, tempfile.TemporaryFile() as fe:
        p = subprocess.Popen(cmd, env=env, stdout=fo, stderr=fe, start_new_session=True)
        timed_out = False
        last = t0
        while True:
            try:
                p.wait(timeout=5.0)
                break
            except subprocess.TimeoutExpired:
                now = time.time()
                if now - last >= 30.0:  # a line now and then: a silent job looks hung to whoever runs it
                    log("... %.0f s" % (now - t0))
                    last = now
                if now - t0 > timeout_s:
                    timed_out = True
                    for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
                        try:
                            os.killpg(p.pid, sig)
                        except ProcessLookupError:
                            pass
                        try:
                            p.wait(timeout=grace)
                            break
                        except subprocess.TimeoutExpired:
                            continue
                    break
        fo.seek(0)
        fe.seek(0)
        out = fo.read().decode("utf-8", "replace")
        err = fe.read().decode("utf-8", "replace")
    return (p.returncode if p.returncode is not None else -9), out, err[-200000:], time.time() - t0, timed_out  # (_err_tail condenses it)


def _err_tail(err, keep=14):
    """the lines of a failed stage's stderr worth keeping: the exception lines of the ranks (`SomeError: message`, the
    first few -- the root cause comes first -- and the last), injected-failure notes, then the end of the stream"""
    import re
    lines = [l for l in err.strip().splitlines() if l.strip()]
    pat = re.compile(r"\b\w*(Error|Exception)\b: \S")
    hits = [l.strip()[:300] for k, l in enumerate(lines)
            if (pat.search(l) or "injected failure" in l or (k and lines[k - 1].strip() == "Last error:"))  # (RCCL's own reason)
            and "ChildFailedError" not in l and "error_file" not in l]
    seen, uniq = set(), []
    for l in hits:
        if l not in seen:
            seen.add(l)
            uniq.append(l)
    head = uniq[:4] + [l for l in uniq[-2:] if l not in uniq[:4]]
    return (head + lines[-max(2, keep - len(head)):])[:keep + 2]


def orchestrate(a, argv):
    """`python bench.py --gpus N` (N > 1) called as a plain script.  Runs the ladder inside `--deadline` seconds, prints
    ONE JSON line -- the winning stage's, with `launcher` saying which stage produced it and what the earlier ones
    died of -- or, when every stage failed, an error line (value null) and a non-zero exit code."""
    t_start = time.time()
    stages = [st for st in (a.ladder.split(",") if a.ladder else LADDER)]
    for st in stages:
        if st not in LADDER:
            raise SystemExit("unknown ladder stage %r (known: %s)" % (st, ", ".join(LADDER)))

    def log(msg):
        print("[bench ladder] " + msg, file=sys.stderr, flush=True)

    failed = []
    for k, stage in enumerate(stages):
        remaining = a.deadline - (time.time() - t_start) - 5.0
        cap = a.stage_timeout if a.stage_timeout > 0 else STAGE_CAP_S[stage]
        # the last stage may use whatever is left; earlier ones leave room for those behind them
        budget = remaining if k == len(stages) - 1 else min(cap, remaining - 45.0 * (len(stages) - 1 - k))
        if a.stage_timeout > 0:
            budget = min(a.stage_timeout, remaining)
        if budget < 15.0:
            failed.append({"stage": stage, "rc": None, "reason": "skipped: %.0f s left of the %.0f s deadline"
                           % (max(remaining, 0.0), a.deadline)})
            continue
        cmd, extra = _stage_cmd(stage, argv, a.gpus)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra)
        log("stage %s (time-out %.0f s)" % (stage, budget))
        rc, out, err, wall, timed_out = _run_stage(cmd, env, budget, log)
        lines = [l for l in out.strip().splitlines() if l.startswith("{")]
        rec = None
        if lines:
            try:
                rec = json.loads(lines[-1])
            except ValueError:
                rec = None
        if rc == 0 and rec is not None and rec.get("value") is not None and "error" not in rec:
            attach_launcher(rec, {"stage": stage, "fallback_from": failed, "stage_wall_s": wall, "ladder": stages,
                                  "deadline_s": a.deadline, "total_wall_s": time.time() - t_start})
            print(json.dumps(rec), flush=True)
            return 0
        reason = ("timed out after %.0f s" % wall) if timed_out else (
            (rec or {}).get("error") or "exit code %d" % rc)
        tail = _err_tail(err)
        failed.append({"stage": stage, "rc": rc, "reason": reason, "wall_s": wall, "stderr_tail": tail})
        log("stage %s failed: %s" % (stage, reason))
        for l in tail:
            log("    " + l[:300])
    print(json.dumps({"metric": METRIC, "value": None, "unit": "GB/s", "n_gpus": a.gpus, "steps": a.steps,
                      "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
                      "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "error": "every stage of the launch ladder failed",
                      "launcher": compact_launcher({"stage": None, "fallback_from": failed, "ladder": stages,
                                                    "deadline_s": a.deadline, "total_wall_s": time.time() - t_start})}),
          flush=True)
    return 1


class RankGuard:
    """One rank of an N-rank job that was NOT started by this file's ladder (the driver launches `python -m
    torch.distributed.run ... bench.py --gpus N` itself): a hang in communicator set-up or a failing rank must still end
    in ONE JSON line.  A watchdog thread per rank:
      * `--rank-deadline` seconds without the job finishing, or an exception in the rank, or another rank's failure note
        (a file keyed by the rendezvous port) -> ranks other than 0 leave QUIETLY with code 0 (a non-zero code would make
        the launcher tear rank 0 down before it can answer); rank 0 waits a moment for their GPUs to be released, then
        runs the rest of the ladder -- `single_process_rccl`, `single_process_fold` -- as FRESH child processes (this
        process has touched the GPU and is never re-executed) and prints the winner's line with `launcher.fallback_from`
        saying what the torch ranks died of, or the error line;
      * SIGTERM from the launcher (some rank crashed hard): rank 0 prints the error line at once."""

    def __init__(self, a, real_stdout):
        import threading
        self.a, self.out = a, real_stdout
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.flag = "/tmp/psp_bench_fail_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "x"))
        self.t0 = time.time()
        self.done = threading.Event()
        self.lock = threading.Lock()
        self.fired = False
        self.thread = threading.Thread(target=self._watch, daemon=True)

    def start(self):
        import signal
        try:
            if self.rank == 0 and os.path.exists(self.flag):
                os.remove(self.flag)
        except OSError:
            pass
        if self.rank == 0:
            # SIGTERM is BLOCKED in this thread (and in every thread started from now on) and picked up by the watchdog with
            # sigtimedwait: a Python-level handler would never run while the main thread sits inside a collective
            signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})
        self.thread.start()

    def _watch(self):
        import signal
        while not self.done.is_set():
            if self.rank == 0:
                if signal.sigtimedwait({signal.SIGTERM}, 2.0) is not None:
                    self._terminated()
            elif self.done.wait(2.0):
                break
            if time.time() - self.t0 > self.a.rank_deadline:
                self.fail("no result after %.0f s (--rank-deadline): a rank hangs" % self.a.rank_deadline)
            if os.path.exists(self.flag):
                try:
                    why = open(self.flag).read()[:300]
                except OSError:
                    why = "another rank failed"
                self.fail(why)

    def _error_line(self, failed, msg):
        a = self.a
        return {"metric": METRIC, "value": None, "unit": "GB/s", "n_gpus": self.world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": None, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
                "data": "synthetic", "error": msg,
                "launcher": compact_launcher({"stage": None, "fallback_from": failed, "ladder": list(LADDER),
                                              "started_by": "external launcher"})}

    def _terminated(self):
        with self.lock:
            if self.fired:
                return
            self.fired = True
        print(json.dumps(self._error_line([{"stage": "torch_rccl_ranks", "rc": None, "reason": "SIGTERM from the launcher "
                                             "(another rank ended abnormally) after %.0f s" % (time.time() - self.t0)}],
                                           "the launcher ended the job")), file=self.out, flush=True)
        os._exit(1)

    def fail(self, reason):
        """called from the watchdog thread or from the rank's own exception handler; never returns"""
        with self.lock:
            if self.fired:
                time.sleep(1e6)
            self.fired = True
        print("[bench rank %d] %s" % (self.rank, reason), file=sys.stderr, flush=True)
        if self.rank != 0:
            try:
                with open(self.flag, "w") as f:
                    f.write("rank %d: %s" % (self.rank, reason))
            except OSError:
                pass
            os._exit(0)
        failed = [{"stage": "torch_rccl_ranks", "rc": None, "reason": reason, "wall_s": time.time() - self.t0}]
        time.sleep(6.0)  # the other ranks see the note / their own deadline and release their GPUs
        argv = [t for t in sys.argv[1:]]
        for stage in LADDER[1:]:
            cmd, extra = _stage_cmd(stage, argv, self.world)
            env = {k: v for k, v in os.environ.items()
                   if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK")}
            env.update(extra)
            budget = self.a.stage_timeout if self.a.stage_timeout > 0 else min(150.0, STAGE_CAP_S[stage])
            rc, out, err, wall, timed_out = _run_stage(cmd, env, budget, lambda m: print("[bench rank 0] " + m, file=sys.stderr, flush=True))
            lines = [l for l in out.strip().splitlines() if l.startswith("{")]
            rec = None
            if lines:
                try:
                    rec = json.loads(lines[-1])
                except ValueError:
                    rec = None
            if rc == 0 and rec is not None and rec.get("value") is not None and "error" not in rec:
                attach_launcher(rec, {"stage": stage, "fallback_from": failed, "stage_wall_s": wall, "ladder": list(LADDER),
                                      "started_by": "external launcher (torch.distributed.run); rank 0 ran the fall-back "
                                                    "stages as fresh child processes"})
                print(json.dumps(rec), file=self.out, flush=True)
                os._exit(0)
            failed.append({"stage": stage, "rc": rc, "reason": ("timed out after %.0f s" % wall) if timed_out else
                           ((rec or {}).get("error") or "exit code %d" % rc), "wall_s": wall, "stderr_tail": _err_tail(err)})
        print(json.dumps(self._error_line(failed, "every stage of the launch ladder failed")), file=self.out, flush=True)
        os._exit(1)

    def finish(self):
        import signal
        self.done.set()
        if self.rank == 0:
            self.thread.join(timeout=5.0)
            signal.pthread_sigmask(signal.SIG_UNBLOCK, {signal.SIGTERM})


def guarded_rank(a, real_stdout, run_body):
    """run_body(a, real_stdout): bench.py's body of one rank"""
    g = RankGuard(a, real_stdout)
    g.start()
    try:
        rc = run_body(a, real_stdout)
    except BaseException as e:  # noqa: BLE001 - whatever the rank died of becomes the reason
        import traceback
        traceback.print_exc()
        g.fail("%s: %s" % (type(e).__name__, str(e)[:300]))
    g.finish()
    return rc
