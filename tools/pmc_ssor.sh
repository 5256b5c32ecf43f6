#!/bin/bash
# tools/pmc_ssor.sh <outdir> : SQ counters of ssor_run_kernel at 2048^2 (no helper workgroups: their sleeping waves would be counted)
OUT=$1; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PSP_TUNING=1 PSP_SSOR_LDS_HELPERS=${HELPERS:-0} PSP_SSOR_GRAPH=0
i=1
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 tools/ssor_timing.py --grid 2048,2048,0 --no-pcg > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
for f in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ssor_run_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(acc.items()):
        print("%-32s %16.1f (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
