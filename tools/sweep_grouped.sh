#!/bin/bash
# Same-box sweep of the grouped decode GEMM configs; variants built with
#   python glm4_moe_nano-vllm_amd/build.py --variant _gN GLM_SM_CFG_GATEUP=a,b,c,d GLM_SM_CFG_DOWN=a,b,c,d
L=$PWD/glm4_moe_nano-vllm_amd/lib
for v in "" "$@"; do
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python bench.py --layers 4 --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 40 2>/dev/null | tail -1 > gpurun_out/sweepg$v.json
  echo "variant '$v'"; python tools/show_bench.py gpurun_out/sweepg$v.json | grep -E "grouped_gemm_(gateup_silu|down) "
done
