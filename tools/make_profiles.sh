#!/bin/bash
# Collect the round's profile artefacts on the GPU box (run via gpurun); everything lands in
# gpurun_out/profiles_rN/ and is then copied by hand into profiles/ (tracked).
#   tools/make_profiles.sh r1
set -u
R=${1:-r1}; OUT=gpurun_out/profiles_$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (1) the bench line itself
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
# (2) kernel trace + stats of the SAME command (no CPU baseline leg: it launches no kernels)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline > $OUT/trace.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/bench_kernel_stats.csv 2>/dev/null
# (3) PMC passes on the SpMV kernel alone (separate passes; FETCH_SIZE and WRITE_SIZE do not fit one)
i=1
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  timeout 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -- python3 tools/prof_spmv.py --reps 3 > $OUT/pmc$i.log 2>&1
  i=$((i+1))
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
vals = {}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "csr_spmv" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kname = [w for w in r["Kernel_Name"].replace("<", " ").replace("(", " ").replace(":", " ").split() if w.startswith("csr_spmv")][0]
    for c, v in acc.items():
        vals[c] = sum(v) / len(v)
with open(os.path.join(out, "spmv_pmc_summary.txt"), "w") as g:
    g.write("# rocprofv3 --pmc averages per launch, %s (default variant), 7-pt Poisson 512^3 (tools/prof_spmv.py)\n" % kname)
    for c in sorted(vals):
        g.write("%-36s %18.1f\n" % (c, vals[c]))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    # MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports exactly half of the bytes of a
    # wide coalesced streaming read on gfx950 -> doubled; WRITE_SIZE (KB) is exact
    hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
    json.dump({"kernel": kname, "workload": "7-pt Poisson 512^3", "FETCH_SIZE_KB": vals["FETCH_SIZE"],
               "WRITE_SIZE_KB": vals["WRITE_SIZE"], "fetch_correction": 2.0,
               "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": 13939769348,
               "note": "L2<->fabric request bytes (Infinity-Cache hits included), not DRAM-only; "
                       "distinct DRAM bytes of csr_spmv_w3 per launch: val 8 + col16 2 per nonzero, "
                       "block list 256 + row offsets 2*64*np per chunk, x and y once"},
              open(os.path.join(out, "spmv_pmc.json"), "w"), indent=1)
PY
cat $OUT/bench.json; head -6 $OUT/bench_kernel_stats.csv | cut -c1-220; cat $OUT/spmv_pmc_summary.txt
