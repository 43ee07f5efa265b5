# fp8 experts at the prefill shape: tiled kernel on the fp8 bytes against expansion pass + bf16 kernel (GLM4MOE_FP8W_NATIVE=0)
timeout -k 10 900 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "fp8" 2>&1 | tail -3 || exit 1
for v in 1 0 1 0; do echo "== GLM4MOE_FP8W_NATIVE=$v"; GLM4MOE_FP8W_NATIVE=$v timeout -k 10 300 python tools/fp8_prefill_probe.py 2>&1 | grep -E "us|rror"; done
