#!/bin/bash
# Same-box timing ablation of the prefill attention kernel (variants built with
# `python glm4_moe_nano-vllm_amd/build.py --variant _faN GLM_FA_ABLATE=N`).  Results of N != 0 are wrong by design.
L=glm4_moe_nano-vllm_amd/lib
for v in "" "$@"; do
  echo -n "variant '$v': "; GLM4MOE_HIP_LIB=$PWD/$L/libglm4moe_hip$v.so python tools/prefill_attn_bench.py
done
