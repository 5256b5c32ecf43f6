#!/bin/bash
# AddressSanitizer + UBSan over the HOST side (GPU sanitizers are not available on this pool): the three CPython extension
# modules (ll_mat feeder, MTX reader threads, conversions, krylov / precon argument handling, the 16-slot C API) and the
# oracle's C are rebuilt with -fsanitize=address,undefined and the CPU test-suite runs on them (host mode children included);
# afterwards everything is rebuilt plain.  Output: gpurun_out/sanitize_host.txt (summary committed under profiles/).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g -O1"
PSP_EXT_CFLAGS="$SAN" python pysparse_amd/build_ext.py --force || exit 1
gcc -O1 -g -fPIC -ffp-contract=off -fvisibility=hidden -std=gnu99 $SAN -shared -o oracle/liboracle.so oracle/pysparse_oracle.c -lm -lpthread || exit 1
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=0:halt_on_error=0 \
  UBSAN_OPTIONS=print_stacktrace=1 \
  timeout 3000 python -m pytest ${@:-tests} -q -m "not gpu" -p no:cacheprovider > gpurun_out/sanitize_host.txt 2>&1
rc=$?
python pysparse_amd/build_ext.py --force
rm -f oracle/liboracle.so && make -s -C oracle liboracle.so
echo "pytest rc=$rc"; tail -5 gpurun_out/sanitize_host.txt
grep -c "ERROR: AddressSanitizer\|runtime error:" gpurun_out/sanitize_host.txt
