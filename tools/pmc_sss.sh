set -u
OUT=gpurun_out/pmc_sss; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
  timeout 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/sss_$i -- python3 tools/prof_spmv.py --reps 3 --sss > $OUT/sss_$i.log 2>&1
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
vals = {}
for f in sorted(glob.glob(os.path.join(out, "sss_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "sss_spmv" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        vals[c] = sum(v) / len(v)
for c in sorted(vals): print("%-36s %18.1f" % (c, vals[c]))
PY
