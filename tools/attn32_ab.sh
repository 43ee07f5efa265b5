mkdir -p gpurun_out/r03_e
python -m pytest tests/test_hip_kernels.py tests/test_fullsize_gpu.py tests/test_layer_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "attn or prefill or engine or layer" > gpurun_out/r03_e/tests_attn.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r03_e/tests_attn.log
for v in 1 0 1 0; do GLM_FA_MFMA32=$v python tools/prefill_attn_bench.py 2>&1 | tail -1 | sed "s/^/MFMA32=$v /"; done | tee gpurun_out/r03_e/attn_ab.txt
