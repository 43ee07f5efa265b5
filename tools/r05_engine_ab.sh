#!/bin/bash
# r05: one-token decode engine — parity test first, then C2 (bs=1 ctx=1024) with and without the engine on the same box.
# usage: bash tools/r05_engine_ab.sh <outdir>
set -o pipefail
OUT=${1:-gpurun_out/r05_a}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -x -q -m gpu -k "decode_mlp_engine or oproj_router_tail or study_kernel_options or qkv_gemm_with_rope" > $OUT/engine_tests.log 2>&1 || { tail -30 $OUT/engine_tests.log; exit 1; }
tail -3 $OUT/engine_tests.log
for v in 0 1 0 1; do
  GLM4MOE_T1_ENGINE=$v timeout -k 10 400 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --kernel-iters 5 2>$OUT/bench_engine$v.err | tail -1 > $OUT/bench_engine$v.json || { tail -20 $OUT/bench_engine$v.err; exit 1; }
  python - <<PY
import json
d=json.load(open("$OUT/bench_engine$v.json"))
c=d["extra"]["c2_bs1_ctx1024"]
print("engine=$v", "C3 tok/s", d["value"], "| C2 tok/s", c["tokens_per_s"], "ms/layer", c["ms_per_layer"], "frac", c["layer_hbm_frac"])
PY
done
