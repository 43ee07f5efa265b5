"""PROTOTYPE measurement: dense bf16 GEMM with a 128 x 128 wave tile (tools/gemm_w4_probe.hip, built by tools/gemm_w4_probe.sh)
against the product's large-M kernel at the prefill projection shapes; bit-identity checked."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
ops.load_library()
lib = ctypes.CDLL(os.path.join(ROOT, "glm4_moe_nano-vllm_amd", "lib", "libw4_probe.so"))
lib.gemm_w4.restype = ctypes.c_int
lib.gemm_w4.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                        ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
for (M, N, K, with_bias) in ((16384, 14336, 4096, True), (16384, 4096, 12288, False), (4096, 4096, 4096, False)):
    x = torch.randn(M, K, device=dev, generator=g, dtype=torch.bfloat16)
    w = ops.padded_weight(N, K, dtype=torch.bfloat16, device=dev).normal_(0.0, K ** -0.5, generator=g)
    b = torch.randn(N, device=dev, generator=g, dtype=torch.bfloat16) if with_bias else None
    y1 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def w4():
        rc = lib.gemm_w4(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), b.data_ptr() if b is not None else None,
                         y1.data_ptr(), N, M, N, K, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc

    y0 = ops.gemm_bf16(x, w, b)
    w4()
    torch.cuda.synchronize()
    same = torch.equal(y0, y1)
    md = float((y0.float() - y1.float()).abs().max())
    fl = 2.0 * M * N * K
    for name, f in (("product", lambda: ops.gemm_bf16(x, w, b)), ("w4 probe", w4), ("product", lambda: ops.gemm_bf16(x, w, b)), ("w4 probe", w4)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(8): f()
        e.record(); torch.cuda.synchronize()
        t = a.elapsed_time(e) / 8 * 1e3
        print(f"M={M} N={N} K={K} {name:9s} {t:9.1f} us  {fl / t / 1e6:7.1f} TFLOP/s  (identical={same}, max diff {md:.3g})", flush=True)
