#!/bin/bash
# Same-box A/B of prefill-attention builds: tools/fa_ab.sh "" _fa_old _fa_nochk   (library suffixes; "" = product)
L=$PWD/glm4_moe_nano-vllm_amd/lib
for round in 1 2 3; do
  for v in "$@"; do
    echo -n "round $round variant '$v': "; GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python tools/prefill_attn_bench.py
  done
done
