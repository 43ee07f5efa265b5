#!/bin/bash
# r05: same-box A/B of the default prefill attention kernel: r04 form (-DGLM_FA_R05=0 -DGLM_FA_STORE16=0, library _r04) |
# product | persistent form (option fa_persist, study library: build.py --study).  bash tools/fa_r05_ab.sh
L=$PWD/glm4_moe_nano-vllm_amd/lib
for shape in "4 4096" "16 1024" "1 16384" "64 256" "8 8192"; do
  for round in 1 2 3; do
    echo -n "shape $shape round $round r04: "; GLM_FA_PERSIST=0 GLM4MOE_HIP_LIB=$L/libglm4moe_hip_r04.so python tools/prefill_attn_bench.py $shape
    echo -n "shape $shape round $round product: "; GLM_FA_PERSIST=0 python tools/prefill_attn_bench.py $shape
    echo -n "shape $shape round $round persist: "; GLM_FA_PERSIST=1 GLM4MOE_HIP_LIB=$L/libglm4moe_hip_study.so python tools/prefill_attn_bench.py $shape
  done
done
