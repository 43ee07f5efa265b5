#!/bin/bash
# Same-box ablation of gemm_smallm: times the hot kernels with each timing-only variant library.
# usage (on the GPU box): bash tools/ablate_smallm.sh
L=glm4_moe_nano-vllm_amd/lib
for v in "" _abl1 _abl2 _abl4 _abl6 _abl8 _abl15; do
  GLM4MOE_HIP_LIB=$PWD/$L/libglm4moe_hip$v.so python bench.py --layers 6 --steps 4 --warmup 1 --no-cpu-baseline --kernel-iters 10 2>/dev/null | tail -1 > gpurun_out/abl$v.json
  echo "variant '$v'"; python tools/show_bench.py gpurun_out/abl$v.json | sed -n 2,6p
done
