#!/bin/bash
# Same-box sweep of the dense decode projection configs (QKV = "mid", O = "small"); variants built with
#   python glm4_moe_nano-vllm_amd/build.py --variant _qN GLM_SM_CFG_MID=a,b,c,d GLM_SM_CFG_SMALL=a,b,c,d
L=$PWD/glm4_moe_nano-vllm_amd/lib
for v in "" "$@"; do
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python bench.py --layers 4 --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 30 2>/dev/null | tail -1 > gpurun_out/sweep$v.json
  echo "variant '$v'"; python tools/show_bench.py gpurun_out/sweep$v.json | grep -E "gemm_qkv|gemm_o"
done
