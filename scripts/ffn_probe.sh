#!/bin/bash
# K7f ablation (GPU box only; the ablation library is built in the container: scripts/microbench/_bin/librectools_hip_abl.so =
# the product objects + rt_ffn.hip compiled with -DRT_ABLATION_BUILD).  Prints the fused kernels' time with parts switched off.
echo "== product library"; timeout 100 python scripts/ffn_bench.py ${1:-13312,16384,18432} 2>&1 | grep "M="
cp scripts/microbench/_bin/librectools_hip_abl.so rectools_amd/librectools_hip.so
for p in 0 1 2 4 7 8 16 32 48 56 63; do echo "== RT_FFN_PROBE=$p"; RT_FFN_PROBE=$p timeout 60 python scripts/ffn_bench.py 13312 fused 2>&1 | grep "M="; done
