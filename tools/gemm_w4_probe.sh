# build the prototype next to the library (git-ignored, travels with the snapshot) — run here, then: gpurun -- python tools/gemm_w4_probe.py
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared tools/gemm_w4_probe.hip -o glm4_moe_nano-vllm_amd/lib/libw4_probe.so
