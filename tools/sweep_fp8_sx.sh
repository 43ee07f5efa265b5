#!/bin/bash
# Same-box sweep of the shared-activation fp8 body's workgroup width (waves per workgroup) for the grouped down / gate|up
# projections at the C3 decode shape.  Build the variants first (CPU box):
#   for n in 2 8; do python glm4_moe_nano-vllm_amd/build.py --variant _sxd$n GLM_SX_DOWN_NW=$n; python glm4_moe_nano-vllm_amd/build.py --variant _sxg$n GLM_SX_GATEUP_NW=$n; done
L=$PWD/glm4_moe_nano-vllm_amd/lib
mkdir -p gpurun_out/sweep_sx
for v in "" _sxd2 _sxd8 _sxg2 _sxg8 "" ; do
  [ -f $L/libglm4moe_hip$v.so ] || continue
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python bench.py --fp8-experts --layers 8 --steps 3 --warmup 1 --no-cpu-baseline --no-c2 --no-extras --kernel-iters 40 2>/dev/null | tail -1 > gpurun_out/sweep_sx/b$v.json
  python - <<PY
import json
d = json.load(open("gpurun_out/sweep_sx/b$v.json"))["kernels"]
print("variant '%s': gate|up %.1f us (%.3f)  down %.1f us (%.3f)" % ("$v", d["grouped_gemm_gateup_silu"]["kernel_us_mean"], d["grouped_gemm_gateup_silu"]["frac"], d["grouped_gemm_down"]["kernel_us_mean"], d["grouped_gemm_down"]["frac"]))
PY
done
