#!/bin/bash
# Prefill attention kernels A/B on one box: default 16x16x32 (0), 32x32x16 (1)
mkdir -p gpurun_out/attn32
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "varlen_attn" > gpurun_out/attn32/tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/attn32/tests.log
for v in 0 1 0 1; do GLM_FA_MFMA32=$v python tools/prefill_attn_bench.py 2>&1 | tail -1 | sed "s/^/MFMA32=$v /"; done | tee gpurun_out/attn32/ab.txt
