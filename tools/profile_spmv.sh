#!/bin/bash
# rocprofv3 passes for the SpMV kernel on one MI355X (run through gpurun).
#   tools/profile_spmv.sh <outdir> <variant> [grid]
# pass 0: kernel trace + stats; passes 1..: PMC counter groups, one run each
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: MI355X_MICROARCH.md).
set -u
OUT=${1:-gpurun_out/prof}; VAR=${2:--1}; GRID=${3:-512,512,512}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 tools/prof_spmv.py --variant $VAR --grid $GRID --reps 5"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE TCC_TAG_STALL_sum TCC_BUBBLE_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -- $CMD > "$OUT/pmc$i.log" 2>&1
  i=$((i+1))
done
# condense: per-kernel averages of every counter for the spmv kernel
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "csr_spmv" in k:
            acc[(k.split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        rows.append((k, c, sum(v) / len(v), len(v)))
with open(os.path.join(out, "pmc_summary.txt"), "w") as g:
    for k, c, m, n in rows:
        g.write("%-64s %-40s %18.1f  (n=%d)\n" % (k, c, m, n))
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    os.system("cp '%s' '%s/kernel_stats.csv'" % (f, out))
PY
ls "$OUT"; cat "$OUT/pmc_summary.txt"; head -8 "$OUT/kernel_stats.csv" 2>/dev/null
