#!/bin/bash
# Round-4 profile artefacts on the GPU box (run via gpurun); everything lands in gpurun_out/profiles_r4/
# and is then copied into profiles/ (tracked).
#   (1) the bench line itself, (2) rocprofv3 --kernel-trace --stats of the SAME command,
#   (3) PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE do not fit one) for csr_spmv_w4 / w3 / w2 on the
#       512^3 operator -> r4_spmv[_w3|_w2]_pmc.json, which bench.py reads for roofline.traffic
set -u
OUT=gpurun_out/profiles_r4; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python3 bench.py > $OUT/r4_bench.json 2> $OUT/bench.err
# the 512^3 legs only: the 1024^3 leg would mix 13 ms launches of the same kernel into its average
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-clocks --no-strong-n1 --no-pmc > $OUT/r4_bench_traced.json 2> $OUT/trace.log
cp $OUT/trace/*/*kernel_stats.csv $OUT/r4_bench_kernel_stats.csv 2>/dev/null
for kv in "w4:-1" "w3:1065154" "w2:16578"; do
  k=${kv%%:*}; v=${kv##*:}; i=1
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS"; do
    timeout 180 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_${k}_$i -- python3 tools/prof_spmv.py --reps 3 --variant $v > $OUT/pmc_${k}_$i.log 2>&1
    i=$((i+1))
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
n, nnz = 134217728, 937951232
for k, fname in (("w4", "r4_spmv"), ("w3", "r4_spmv_w3"), ("w2", "r4_spmv_w2")):
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
# (4) the other measurements quoted in DESIGN.md
rm -f $OUT/r4_fem_standin.txt
for s in 1 32 512; do timeout 300 python3 tools/fem_standin.py --shuffle $s --variants 16513 >> $OUT/r4_fem_standin.txt 2>> $OUT/tools.err; done
# the same stand-in with rows that couple to unknowns anywhere (outlier chunks / hubs)
for sw in "1 200" "1 4000" "512 200"; do set -- $sw; timeout 300 python3 tools/fem_standin.py --shuffle $1 --wild $2 >> $OUT/r4_fem_standin.txt 2>> $OUT/tools.err; done
timeout 300 python3 tools/small_solver_timing.py > $OUT/r4_small_solvers.txt 2>> $OUT/tools.err
timeout 300 python3 tools/minres_timing.py > $OUT/r4_minres_timing.txt 2>> $OUT/tools.err
timeout 300 python3 tools/bench_configs.py > $OUT/r4_configs.json 2>> $OUT/tools.err
timeout 600 python3 bench.py --gpus 1 --scaling strong --no-cpu-baseline > $OUT/r4_bench_strong_world1.json 2>> $OUT/tools.err
ls $OUT | head -60; cat $OUT/r4_bench.json | head -c 1500; echo; head -8 $OUT/r4_bench_kernel_stats.csv | cut -c1-160
timeout 600 python3 tools/extra_solver_timing.py > $OUT/r4_extra_solvers.txt 2>> $OUT/tools.err
# (5) round 3+4: the host-pointer product against the PCIe link, the one-process device-list path (4 ranks sharing this GPU at
#     configs[3]'s true size: a rehearsal, not a measurement), SSOR
timeout 600 python3 tools/host_matvec_timing.py > $OUT/r4_host_matvec.json 2>> $OUT/tools.err
timeout 900 python3 bench.py --gpus 4 --single-process --share-gpu --steps 10 --warmup 3 --pcg-iters 16 > $OUT/r4_single_process_n4_1024_one_gpu.json 2>> $OUT/tools.err
# (6) round 4: the launch ladder end to end on one GPU -- torch ranks over gloo sharing cuda:0 (stage 1), the same with a rank that
#     dies (falls through to the single-process stage); configs[4] through bench.py --mtx on the three stand-ins
timeout 900 python3 bench.py --gpus 3 --backend gloo --share-gpu --grid 256,256,255 --steps 10 --warmup 3 --pcg-iters 40 --no-cpu-baseline --no-clocks > $OUT/r4_ladder_3ranks_gloo_one_gpu.json 2>> $OUT/tools.err
timeout 900 python3 bench.py --gpus 3 --backend gloo --share-gpu --grid 256,256,255 --steps 10 --warmup 3 --pcg-iters 40 --no-cpu-baseline --no-clocks --inject exit:1 > $OUT/r4_ladder_fallback_one_gpu.json 2>> $OUT/tools.err
# eight ranks (the node's rank count) in ONE process sharing this GPU at configs[3]'s true size, and four torch ranks over gloo sharing it (the box allows six processes on its GPU at once: 6 ranks trip the guard)
timeout 900 python3 bench.py --gpus 8 --single-process --share-gpu --steps 10 --warmup 3 --pcg-iters 16 > $OUT/r4_single_process_n8_1024_one_gpu.json 2>> $OUT/tools.err
timeout 900 python3 bench.py --gpus 4 --backend gloo --share-gpu --grid 256,256,252 --steps 6 --warmup 2 --pcg-iters 30 --no-cpu-baseline --no-clocks --ladder torch_rccl_ranks > $OUT/r4_ladder_4ranks_gloo_one_gpu.json 2>> $OUT/tools.err
rm -f $OUT/r4_mtx_leg_standins.jsonl
for s in fem32 fem512 logspaced; do timeout 300 python3 bench.py --mtx standin:$s >> $OUT/r4_mtx_leg_standins.jsonl 2>> $OUT/tools.err; done
timeout 600 python3 bench.py --gpus 1 --single-process --steps 20 --warmup 5 --pcg-iters 32 > $OUT/r4_single_process_n1_512.json 2>> $OUT/tools.err
tail -3 $OUT/tools.err
