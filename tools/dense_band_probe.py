"""L2 band height of the large-M GEMM for the DENSE prefill projections (QKV 16384x14336x4096, o_proj 16384x4096x12288):
glm_debug_largem_band switches it between launches of one process."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
# the variant / band switches exist only in the study build of the library:
#     python glm4_moe_nano-vllm_amd/build.py --variant _study GLM_LM_STUDY=1
_STUDY = os.path.join(ROOT, "glm4_moe_nano-vllm_amd", "lib", "libglm4moe_hip_study.so")
os.environ.setdefault("GLM4MOE_HIP_LIB", _STUDY)
if not os.path.exists(os.environ["GLM4MOE_HIP_LIB"]):
    raise SystemExit("build the study library first: python glm4_moe_nano-vllm_amd/build.py --variant _study GLM_LM_STUDY=1")
import ctypes
from nanovllm import hip_ops as ops
lib = ops.load_library()
lib.glm_debug_largem_band.restype, lib.glm_debug_largem_band.argtypes = ctypes.c_int, [ctypes.c_int]
T = 16384
shapes = {"qkv": (14336, 4096), "o_proj": (4096, 12288)}
g = torch.Generator(device="cuda").manual_seed(3)
for name, (N, K) in shapes.items():
    x = torch.randn(T, K, device="cuda", generator=g).to(torch.bfloat16)
    w = ops.padded_weight(N, K, dtype=torch.bfloat16, device="cuda")
    w.normal_(0.0, 0.02, generator=g)
    fl = 2.0 * T * N * K
    for _ in range(3): ops.gemm_bf16(x, w, None)
    res = {}
    for rnd in range(3):
        for band in (1, 2, 4, 8, 16):
            lib.glm_debug_largem_band(band)
            ops.gemm_bf16(x, w, None)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): ops.gemm_bf16(x, w, None)
            b.record(); torch.cuda.synchronize()
            res.setdefault(band, []).append(a.elapsed_time(b) / 5 * 1e3)
    lib.glm_debug_largem_band(4)
    print(name, {band: f"{sorted(v)[1]:.0f} us = {fl / sorted(v)[1] / 1e6:.0f} TFLOP/s" for band, v in res.items()}, flush=True)
