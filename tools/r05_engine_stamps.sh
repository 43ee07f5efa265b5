#!/bin/bash
set -o pipefail
OUT=${1:-gpurun_out/r05_b}
mkdir -p $OUT
GLM4MOE_HIP_LIB=$PWD/glm4_moe_nano-vllm_amd/lib/libglm4moe_hip_enstamp.so timeout -k 10 300 python tools/engine_stamps.py > $OUT/engine_stamps.txt 2>&1 || { tail -20 $OUT/engine_stamps.txt; exit 1; }
cat $OUT/engine_stamps.txt
