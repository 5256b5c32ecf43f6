#!/bin/bash
# tools/pmc_fem.sh <outdir> <shuffle>: counters of the kernels of the irregular stand-in (csr_spmv_w3 on the renumbered copy,
# the two permutation passes), one rocprofv3 --pmc pass per group
OUT=$1; SH=$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_LDS" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum"; do
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc$i -- python3 tools/fem_standin.py --shuffle $SH > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        for key in ("csr_spmv_w3", "permute_gather", "permute_back", "csr_spmv_w5", "csr_spmv_w2"):
            if key in name:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        for c, v in sorted(acc[k].items()):
            print("%-18s %-34s %16.1f (n=%d)" % (k, c, sum(v) / len(v), len(v)))
PY
