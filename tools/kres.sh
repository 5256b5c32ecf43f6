#!/bin/bash
# kernel resource usage of one .hip file: name, VGPRs, SGPRs, occupancy (waves/SIMD), LDS, scratch
hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -Iinclude -Ipysparse_amd/csrc -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep "remark:" | sed -E 's/.*remark: +//; s/ \[-Rpass.*//' \
 | awk '/^Function Name/{if(n)print n,v,s,o,l,sc; n=$3} /^VGPRs:/{v="vgpr="$2} /^TotalSGPRs/{s="sgpr="$2} /^Occupancy/{o="occ="$3} /^LDS Size/{l="lds="$4} /^ScratchSize/{sc="scratch="$3} END{print n,v,s,o,l,sc}' | c++filt | sed -E 's/\(anonymous namespace\):://; s/\(.*\)//' 
