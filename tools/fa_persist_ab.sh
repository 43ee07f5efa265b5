#!/bin/bash
# r05: same-box A/B of the persistent prefill attention (option fa_persist) against the one-block-per-workgroup kernel.
#   bash tools/fa_persist_ab.sh [lib suffixes for the persistent runs, default ""]
L=$PWD/glm4_moe_nano-vllm_amd/lib
[ $# -eq 0 ] && set -- ""
for shape in "4 4096" "16 1024" "1 16384" "64 256"; do
  for round in 1 2 3; do
    echo -n "shape $shape round $round default (8-byte stores): "; GLM_FA_PERSIST=0 GLM4MOE_HIP_LIB=$L/libglm4moe_hip_st8.so python tools/prefill_attn_bench.py $shape
    echo -n "shape $shape round $round default: "; GLM_FA_PERSIST=0 python tools/prefill_attn_bench.py $shape
    for v in "$@"; do
      echo -n "shape $shape round $round persist '$v': "; GLM_FA_PERSIST=1 GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python tools/prefill_attn_bench.py $shape
    done
  done
done
