#!/bin/bash
# A/B two builds of the library on the PCG iteration rate (same GPU, alternating processes)
for i in 1 2 3; do
  for lib in pysparse_amd/libpysparse_hip.so build/libpysparse_hip_nt.so; do
    cp $lib /tmp/lib_ab.so
    PSP_LIB_OVERRIDE=/tmp/lib_ab.so python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['pcg_iters_per_s'],1))"
  done
done
