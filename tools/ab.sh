#!/bin/bash
# same-box A/B: bash tools/ab.sh <libsuffixA> <libsuffixB> [layers]
L=$PWD/glm4_moe_nano-vllm_amd/lib
for r in 1 2; do for v in "$1" "$2"; do
  GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python bench.py --layers ${3:-6} --steps 4 --warmup 1 --no-cpu-baseline --kernel-iters 10 2>/dev/null | tail -1 > gpurun_out/ab_$r$v.json
  echo "round $r variant '$v'"; python tools/show_bench.py gpurun_out/ab_$r$v.json | sed -n 1,6p
done; done
