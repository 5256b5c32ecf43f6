#!/usr/bin/env python3
"""Which hardware counter differs between the two timing modes of a process?

Every pass is a FRESH process of tools/prof_spmv.py (512^3, csr_spmv_w4 + a few PCG iterations) under
`rocprofv3 --pmc <group>`; the pass's own kernel durations say which mode that process landed in, the counters
of the same dispatches say what was different.  Usage (on the GPU box):
    python3 tools/mode_census.py gpurun_out/modes [rounds]
Writes <out>/census.jsonl (one line per pass) and prints a table; tools/mode_census.py --summarise <files...>
merges the jsonl files of several gpurun calls (the mode goes mostly with the box, so one call rarely sees both)."""
import collections
import csv
import glob
import json
import os
import re
import subprocess
import sys
import time

WANT = [
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RD_UNCACHED_32B_sum"],
    ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_WRREQ_STALL_sum"],
    ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum"],
    ["TCC_TAG_STALL_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum", "TCC_BUBBLE_sum"],
    ["TCC_EA0_WRREQ_IO_CREDIT_STALL_sum", "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum",
     "TCC_EA0_WR_UNCACHED_32B_sum"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_WRITEBACK_sum", "TCC_NORMAL_WRITEBACK_sum"],
    ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_PERMISSION_MISS_sum",
     "TCP_PENDING_STALL_CYCLES_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAVES"],
    ["TCC_CYCLE_sum", "TCC_BUSY_sum", "TCC_REQ_sum", "TCC_STREAMING_REQ_sum"],
    ["TCC_EA0_RDREQ", "TCC_EA0_WRREQ"],  # unsummed: one value per TCC instance when the tool reports dimensions
    # 10: clocks of three blocks + memory-side occupancy in ONE process (GRBM, SQ and TCC slots are independent)
    ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "TCC_CYCLE_sum", "TCC_BUSY_sum",
     "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_WRREQ_LEVEL_sum"],
]
KERNELS = ("csr_spmv_w4", "px_update", "r_update", "pupdate", "x_update")


def available():
    try:
        out = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120).stdout
    except Exception as e:  # noqa: BLE001
        print("rocprofv3 -L failed:", e)
        return set(), ""
    return set(re.findall(r"\b[A-Z][A-Za-z0-9_]{3,}\b", out)), out


def one_pass(out, idx, group):
    d = os.path.join(out, "pass%03d" % idx)
    os.makedirs(d, exist_ok=True)
    cmd = ["rocprofv3", "--pmc"] + group + ["--output-format", "csv", "-d", d, "--", sys.executable,
                                             "tools/prof_spmv.py", "--reps", "6", "--pcg", "4"]
    t0 = time.time()
    with open(d + ".log", "w") as lg:
        rc = subprocess.run(cmd, stdout=lg, stderr=subprocess.STDOUT, timeout=300).returncode
    rec = {"pass": idx, "group": group, "rc": rc, "wall_s": round(time.time() - t0, 1), "kernels": {}}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: {"dur_ms": [], "ctr": collections.defaultdict(list)})
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            k = next((k for k in KERNELS if k in name), None)
            if k is None:
                continue
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                per[k]["dur_ms"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
            per[k]["ctr"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            durs = sorted(v["dur_ms"])
            rec["kernels"][k] = {"n": len(durs), "ms_med": durs[len(durs) // 2], "ms_min": durs[0],
                                 "ctr": {c: sum(x) / len(durs) for c, x in v["ctr"].items()}}
    return rec


def fmt(rec):
    k = rec["kernels"].get("csr_spmv_w4")
    if not k:
        return "pass %3d rc %d (no csr_spmv_w4 rows) %s" % (rec["pass"], rec["rc"], rec["group"])
    return "pass %3d w4 %.4f ms | " % (rec["pass"], k["ms_med"]) + " ".join(
        "%s=%.6g" % (c.replace("TCC_EA0_", "EA_").replace("_sum", ""), v) for c, v in sorted(k["ctr"].items()))


def summarise(files):
    recs = [json.loads(l) for f in files for l in open(f) if l.strip()]
    rows = collections.defaultdict(list)
    for r in recs:
        for kname, k in r["kernels"].items():
            for c, v in k["ctr"].items():
                rows[(kname, c)].append((k["ms_med"], v, k["n"]))
    print("%-14s %-40s %s" % ("kernel", "counter", "(ms, value per launch) by pass, sorted by time"))
    for (kname, c), vals in sorted(rows.items()):
        vals.sort()
        print("%-14s %-40s %s" % (kname, c, "  ".join("%.3f:%.5g" % (m, v) for m, v, _ in vals)))


def main():
    if sys.argv[1] == "--summarise":
        return summarise(sys.argv[2:])
    out = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    os.makedirs(out, exist_ok=True)
    names, raw = available()
    with open(os.path.join(out, "rocprofv3_L.txt"), "w") as f:
        f.write(raw)
    groups = [[c for c in g if c in names] for g in WANT]
    groups = [g for g in groups if g]
    if os.environ.get("CENSUS_GROUPS"):  # e.g. "10" or "2,10": indices into WANT
        groups = [groups[int(i)] for i in os.environ["CENSUS_GROUPS"].split(",")]
    print("counter groups:", groups, flush=True)
    idx = 0
    with open(os.path.join(out, "census.jsonl"), "a") as jf:
        for _ in range(rounds):
            for g in groups:
                rec = one_pass(out, idx, g)
                jf.write(json.dumps(rec) + "\n")
                jf.flush()
                print(fmt(rec), flush=True)
                idx += 1


if __name__ == "__main__":
    main()
