#!/bin/bash
# same-box A/B of prefill GEMM variants: bash tools/ab_prefill.sh "" _band4 _band16 ...
L=$PWD/glm4_moe_nano-vllm_amd/lib
for r in 1 2; do for v in "$@"; do echo "round $r variant '$v'"; GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python tools/prefill_gemm_only.py 2>/dev/null; done; done
