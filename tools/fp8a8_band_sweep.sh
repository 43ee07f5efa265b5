#!/bin/bash
# r05: row blocks per L2 band (GLM_LM_BAND) for the fp8 x fp8 pair — variant libraries lib/libglm4moe_hip_band{2,8,16}.so
# (python glm4_moe_nano-vllm_amd/build.py --variant _band8 GLM_LM_BAND=8), product library = 4
OUT=${1:-gpurun_out/r05_band}; mkdir -p $OUT
L=$PWD/glm4_moe_nano-vllm_amd/lib
for v in "" _band2 _band8 _band16 ""; do
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so timeout -k 10 200 python tools/fp8a8_probe.py > $OUT/probe$v.txt 2>&1
  echo "== band variant '$v'"; grep -E "^(bf16|fp8_w8a8)" -A3 $OUT/probe$v.txt | grep -E "grouped_gemm@|^bf16|^fp8_w8a8"
done
