#!/bin/bash
# Round-5 profile artefacts on the GPU box (run via gpurun in two calls: `make_profiles_r5.sh a`, then `b`); everything lands
# in gpurun_out/profiles_r5/ and is then copied into profiles/ (tracked).
#   a: (1) the bench line itself, (2) rocprofv3 --kernel-trace --stats of the SAME command, (3) PMC passes (separate runs;
#      FETCH_SIZE and WRITE_SIZE do not fit one) for csr_spmv_w4 / w3 / w6 / w2 on the 512^3 operator -> r5_spmv*_pmc.json,
#      which bench.py falls back on for roofline.traffic where counters cannot be read in the job
#   b: the other measurements quoted in DESIGN.md (configs, the six solvers, stand-ins of configs[4], ladder rehearsals)
set -u
OUT=gpurun_out/profiles_r5; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "${1:-a}" = "a" ]; then
timeout 600 python3 bench.py > $OUT/r5_bench.json 2> $OUT/bench.err
# the 512^3 legs only: the 1024^3 leg would mix 13 ms launches of the same kernel into its average
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-clocks --no-strong-n1 --no-pmc --no-solvers > $OUT/r5_bench_traced.json 2> $OUT/trace.log
cp $OUT/trace/*/*kernel_stats.csv $OUT/r5_bench_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace
for kv in "w4:-1" "w3:1065154" "w6:8405186" "w2:16578"; do
  k=${kv%%:*}; v=${kv##*:}; i=1
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"; do
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${k}_$i -- python3 tools/prof_spmv.py --reps 3 --variant $v > $OUT/pmc_${k}_$i.log 2>&1
    i=$((i+1))
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
n, nnz = 134217728, 937951232
for k, fname in (("w4", "r5_spmv"), ("w3", "r5_spmv_w3"), ("w6", "r5_spmv_w6"), ("w2", "r5_spmv_w2")):
    vals, kname = {}, None
    for f in sorted(glob.glob(os.path.join(out, "pmc_%s_*" % k, "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "csr_spmv" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
                kname = [w for w in r["Kernel_Name"].replace("<", " ").replace("(", " ").replace(":", " ").split() if w.startswith("csr_spmv")][0]
        for c, v in acc.items():
            vals[c] = sum(v) / len(v)
    with open(os.path.join(out, fname + "_pmc_summary.txt"), "w") as g:
        g.write("# rocprofv3 --pmc averages per launch, %s, 7-pt Poisson 512^3 (tools/prof_spmv.py)\n" % kname)
        for c in sorted(vals):
            g.write("%-36s %18.1f\n" % (c, vals[c]))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        # MI355X_MICROARCH.md section HBM: FETCH_SIZE (KB) reports exactly half of the bytes of a wide coalesced
        # streaming read on gfx950 -> doubled; WRITE_SIZE (KB) is exact for 16-byte-per-lane streaming stores
        hbm = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        json.dump({"kernel": kname, "workload": "7-pt Poisson 512^3", "FETCH_SIZE_KB": vals["FETCH_SIZE"],
                   "WRITE_SIZE_KB": vals["WRITE_SIZE"], "fetch_correction": 2.0, "hbm_bytes_per_launch": hbm,
                   "csr_model_bytes_per_launch": 12 * nnz + 20 * n + 4,
                   "note": "L2<->fabric request bytes (Infinity-Cache hits are counted, MI355X_MICROARCH.md), "
                           "not DRAM-only"},
                  open(os.path.join(out, fname + "_pmc.json"), "w"), indent=1)
PY
rm -rf $OUT/pmc_*_[0-9]
ls $OUT; head -c 600 $OUT/r5_bench.json; echo; head -6 $OUT/r5_bench_kernel_stats.csv | cut -c1-200
else
timeout 300 python3 tools/bench_configs.py > $OUT/r5_configs.json 2>> $OUT/tools.err
timeout 300 python3 tools/small_solver_timing.py > $OUT/r5_small_solvers.txt 2>> $OUT/tools.err
timeout 300 python3 tools/minres_timing.py > $OUT/r5_minres_timing.txt 2>> $OUT/tools.err
timeout 600 python3 tools/extra_solver_timing.py > $OUT/r5_extra_solvers.txt 2>> $OUT/tools.err
timeout 300 python3 tools/extra_solver_timing.py --grid 1024,1024,0 --short 100 --long 1100 > $OUT/r5_extra_solvers_1024sq.txt 2>> $OUT/tools.err
timeout 300 python3 tools/extra_solver_timing.py --grid 4096,4096,0 --short 20 --long 120 > $OUT/r5_extra_solvers_c2.txt 2>> $OUT/tools.err
rm -f $OUT/r5_mtx_leg_standins.jsonl
for s in fem32 fem512 logspaced; do timeout 300 python3 bench.py --mtx standin:$s >> $OUT/r5_mtx_leg_standins.jsonl 2>> $OUT/tools.err; done
timeout 600 python3 bench.py --gpus 1 --scaling strong --no-cpu-baseline > $OUT/r5_bench_strong_world1.json 2>> $OUT/tools.err
timeout 900 python3 bench.py --gpus 3 --backend gloo --share-gpu --grid 256,256,255 --steps 10 --warmup 3 --pcg-iters 40 --no-cpu-baseline --no-clocks > $OUT/r5_ladder_3ranks_gloo_one_gpu.json 2>> $OUT/tools.err
timeout 900 python3 bench.py --gpus 4 --single-process --share-gpu --steps 10 --warmup 3 --pcg-iters 16 > $OUT/r5_single_process_n4_1024_one_gpu.json 2>> $OUT/tools.err
ls -la $OUT; tail -3 $OUT/tools.err
fi
