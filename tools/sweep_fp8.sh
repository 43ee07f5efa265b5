#!/bin/bash
# Same-box sweep of the fp8-weight grouped GEMM configs (variants: GLM_SM_CFG_GATEUP_W8 / GLM_SM_CFG_DOWN_W8)
L=$PWD/glm4_moe_nano-vllm_amd/lib
for v in "" "$@"; do
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python bench.py --fp8-experts --layers 4 --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 40 2>/dev/null | tail -1 > gpurun_out/sweepw$v.json
  echo "variant '$v'"; python tools/show_bench.py gpurun_out/sweepw$v.json | grep -E "grouped_gemm_(gateup_silu|down) "
done
