#!/bin/bash
# correlate per-process SpMV time with TLB translation misses (tuning aid)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  OUT=gpurun_out/tlb/run$i; mkdir -p $OUT
  timeout 120 rocprofv3 --pmc ${PMC:-GRBM_GUI_ACTIVE SQ_BUSY_CYCLES} --output-format csv -d $OUT -- python3 tools/prof_spmv.py --reps 4 ${EXTRA:-} > $OUT.log 2>&1
  python3 - $OUT <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list); dur = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "csr_spmv" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
d = sorted(set(dur))
print("ms %s | " % " ".join("%.3f" % x for x in d[:4]) + " ".join("%s=%.3g" % (k.replace("TCP_UTCL1_", ""), sum(v) / len(v)) for k, v in sorted(acc.items())))
PY
done
