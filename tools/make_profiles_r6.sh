#!/bin/bash
# Round-6 profile artefacts on the GPU box (one gpurun call); everything lands in gpurun_out/profiles_r6/ and is then copied
# into profiles/ (tracked):  (1) the default bench line and its side file, (2) rocprofv3 --kernel-trace --stats of the SAME
# command (512^3 legs only: the 1024^3 leg would mix 13 ms launches of the same kernel into its average), (3) PMC passes
# (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one) for csr_spmv_w4 / w6 on the 512^3 operator, (4) the N > 1 code
# paths rehearsed on this one GPU with the round-6 line (predicted / missed_budget beside phases)
set -u
OUT=gpurun_out/profiles_r6; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 bench.py --side-file $OUT/r6_bench_side.json > $OUT/r6_bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-clocks --no-strong-n1 --no-pmc --no-solvers --side-file $OUT/r6_bench_traced_side.json > $OUT/r6_bench_traced.json 2> $OUT/trace.log
cp $OUT/trace/*/*kernel_stats.csv $OUT/r6_bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
for kv in "w4:-1" "w6:8405186"; do
  k=${kv%%:*}; v=${kv##*:}; i=1
  for grp in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${k}_$i -- python3 tools/prof_spmv.py --reps 3 --variant $v > $OUT/pmc_${k}_$i.log 2>&1
    i=$((i+1))
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
n, nnz = 134217728, 937951232
for k, fname in (("w4", "r6_spmv"), ("w6", "r6_spmv_w6")):
    vals, kname = {}, None
    for f in sorted(glob.glob(os.path.join(out, "pmc_%s_*" % k, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "csr_spmv" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kname = [w for w in r["Kernel_Name"].replace("<", " ").replace("(", " ").replace(":", " ").split() if w.startswith("csr_spmv")][0]
        for c, v in acc.items():
            vals[c] = sum(v) / len(v)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports exactly half of the bytes of a wide coalesced
        # streaming read on gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-byte-per-lane streaming stores
        hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        json.dump({"kernel": kname, "workload": "7-pt Poisson 512^3", "FETCH_SIZE_KB": vals["FETCH_SIZE"],
                   "WRITE_SIZE_KB": vals["WRITE_SIZE"], "fetch_correction": 2.0, "hbm_bytes_per_launch": hbm,
                   "csr_model_bytes_per_launch": 12 * nnz + 20 * n + 4,
                   "note": "L2<->fabric request bytes (Infinity-Cache hits are counted, MI355X_MICROARCH.md), not DRAM-only"},
                  open(os.path.join(out, fname + "_pmc.json"), "w"), indent=1)
PY
rm -rf $OUT/pmc_*_[0-9]
timeout 900 python3 bench.py --gpus 3 --backend gloo --share-gpu --grid 256,256,255 --steps 10 --warmup 3 --pcg-iters 40 --no-cpu-baseline --no-clocks --side-file $OUT/r6_ladder_3ranks_gloo_one_gpu_side.json > $OUT/r6_ladder_3ranks_gloo_one_gpu.json 2>> $OUT/tools.err
timeout 900 python3 bench.py --gpus 4 --single-process --share-gpu --steps 10 --warmup 3 --pcg-iters 16 --side-file $OUT/r6_single_process_n4_1024_one_gpu_side.json > $OUT/r6_single_process_n4_1024_one_gpu.json 2>> $OUT/tools.err
ls -la $OUT; head -c 400 $OUT/r6_bench.json; echo; head -5 $OUT/r6_bench_kernel_stats.csv | cut -c1-220
