#!/bin/bash
# tools/kry_floor.py for the four solvers at GRID (default 1024,1024,0) -> gpurun_out/r5_kry_floor.txt (run on the GPU box)
set -e
GRID=${1:-1024,1024,0}
OUT=gpurun_out/r5_kry_floor.txt
: > $OUT
for s in cgs bicgstab qmrs gmres; do
  it=400
  us=$(python3 tools/kry_floor.py time $s $GRID | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['us_per_iter'])")
  rm -rf gpurun_out/kry_$s
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kry_$s -- python3 tools/kry_floor.py run $s $GRID $it > /dev/null 2>&1
  echo "== $s at $GRID: $us us per iteration (two truncated solves, no profiler)" >> $OUT
  python3 tools/kry_floor.py sum gpurun_out/kry_$s $it $us >> $OUT
  rm -rf gpurun_out/kry_$s
done
cat $OUT
