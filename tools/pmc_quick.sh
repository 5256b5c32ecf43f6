#!/bin/bash
# tools/pmc_quick.sh <outdir> <variant> : FETCH/WRITE/TCC hit counters for one spmv variant (each pass under timeout)
OUT=$1; VAR=$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 tools/prof_spmv.py --variant $VAR --reps 3 > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "csr_spmv" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print("%-24s %16.1f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
