#!/bin/bash
# Round-5 counter passes on the GPU box (gpurun): csr_spmv_w6 beside w2 at 512^3, and the PCG product with / without the
# folded p + x updates.  Each --pmc group is its own run (MI355X_MICROARCH.md); outputs under gpurun_out/pmc_r5/.
set -u
OUT=gpurun_out/pmc_r5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
W2=16578; W6=8405186
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  for kv in "w2:$W2" "w6:$W6"; do
    k=${kv%%:*}; v=${kv##*:}
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/${k}_$i -- python3 tools/prof_spmv.py --reps 3 --variant $v > $OUT/${k}_$i.log 2>&1
  done
  i=$((i+1))
done
export PSP_TUNING=1
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  for pf in 0 1; do
    export PSP_PCG_LAZYPF=$pf
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/pcg_pf${pf}_$i -- python3 tools/prof_spmv.py --reps 1 --pcg 12 > $OUT/pcg_pf${pf}_$i.log 2>&1
  done
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
def collect(prefix, match):
    vals = {}
    for f in sorted(glob.glob(os.path.join(out, prefix + "_*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            kn = r.get("Kernel_Name", "")
            for m in match:
                if m in kn:
                    acc[m][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for m, cs in acc.items():
            for c, v in cs.items():
                vals[(m, c)] = (sum(v) / len(v), len(v))
    return vals
with open(os.path.join(out, "summary.txt"), "w") as g:
    for prefix, match in (("w2", ["csr_spmv_w2"]), ("w6", ["csr_spmv_w6"]), ("pcg_pf0", ["csr_spmv_w4", "px_update_kernel", "r_update_kernel"]),
                          ("pcg_pf1", ["csr_spmv_w4_pf", "r_update_kernel"])):
        vals = collect(prefix, match)
        g.write("# %s (rocprofv3 --pmc, averages per launch; FETCH_SIZE / WRITE_SIZE in KB, FETCH_SIZE to be doubled per MI355X_MICROARCH.md)\n" % prefix)
        for (m, c), (v, cnt) in sorted(vals.items()):
            g.write("%-22s %-34s %18.1f  (%d launches)\n" % (m, c, v, cnt))
print(open(os.path.join(out, "summary.txt")).read())
PY
