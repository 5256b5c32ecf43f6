#!/bin/bash
# tools/pmc_any.sh <outdir> <kernel-substring> <prof_spmv.py args...> : FETCH/WRITE/TCC counters, one pass each
OUT=$1; KSUB=$2; shift 2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU"; do
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 tools/prof_spmv.py --reps 3 "$@" > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python3 - $OUT $KSUB <<'PY'
import csv, glob, os, sys, collections
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print("%-32s %16.1f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
