#!/bin/bash
# grouped decode GEMM configs across batch sizes (variant libs as in tools/sweep_grouped.sh)
L=$PWD/glm4_moe_nano-vllm_amd/lib
for b in 1 2 4 8 16; do for v in "" "$@"; do
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python bench.py --batch $b --ctx 1024 --layers 4 --steps 3 --warmup 1 --no-cpu-baseline --kernel-iters 30 2>/dev/null | tail -1 > gpurun_out/sweepb$v.json
  echo "batch $b variant '$v': $(python tools/show_bench.py gpurun_out/sweepb$v.json | grep -E 'grouped_gemm_(gateup_silu|down) ' | sed -E 's/ +/ /g' | cut -d' ' -f2,7,8 | tr '\n' ' ')"
done; done
