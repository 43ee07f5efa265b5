#!/bin/bash
# bs=1 ctx=1024 (BASELINE config 2) hot-kernel times for variant libraries (suffixes as arguments), one bench run each
L=$PWD/glm4_moe_nano-vllm_amd/lib
for v in "" "$@"; do
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python bench.py --batch 1 --ctx 1024 --layers 6 --steps 5 --warmup 2 --no-cpu-baseline --kernel-iters 40 2>/dev/null | tail -1 > gpurun_out/sweepc2$v.json
  echo "variant '$v': $(python tools/show_bench.py gpurun_out/sweepc2$v.json | grep -E ' (grouped_gemm_gateup_silu|grouped_gemm_down|gemm_qkv|gemm_o|paged_attn_decode) ' | sed -E 's/ +/ /g' | cut -d' ' -f2,7 | tr '\n' ' ')"
done
