L=$PWD/glm4_moe_nano-vllm_amd/lib
for v in "" _g4 _d2 "" _g4 _d2; do echo "== variant '$v'"; GLM4MOE_HIP_LIB=$L/libglm4moe_hip$v.so python tools/tinym_probe.py 2>/dev/null | grep "T=1" | grep -E "gateup|down"; done
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "tinym" 2>&1 | tail -2
