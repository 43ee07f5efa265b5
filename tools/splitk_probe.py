"""Time the K-split partial-sum GEMM (glm_gemm_bf16_partial) against glm_gemm_bf16 at the decode O / QKV shapes."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
lib = ops.load_library()
f = lib.glm_gemm_bf16_partial
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
              ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sl = lib.glm_gemm_bf16_partial_slices; sl.restype = ctypes.c_int; sl.argtypes = [ctypes.c_int, ctypes.c_int]
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for name, M, N, K in (("O", 32, 4096, 12288), ("QKV", 32, 14336, 4096)):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    ws = []
    for i in range(6):  # rotate weights so that nothing stays in the caches
        w = ops.padded_weight(N, K, dtype=torch.bfloat16, device="cuda"); w.copy_(torch.randn(N, K, device="cuda") * K ** -0.5); ws.append(w)
    ref = ops.gemm_bf16(x, ws[0], None).float()
    t0 = timeit(lambda i=0: ops.gemm_bf16(x, ws[i % 6], None))
    print(f"{name}: gemm_bf16 {t0:.1f} us")
    for want in (2, 3, 4, 6, 8, 12, 16):
        ns = sl(K, want)
        part = torch.empty(ns, M, N, dtype=torch.float32, device="cuda")
        def run(i=0):
            w = ws[i % 6]
            rc = f(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), part.data_ptr(), M, N, K, ns, None)
            assert rc == 0, ops.last_error() if hasattr(ops, "last_error") else rc
        run(0); torch.cuda.synchronize()
        got = part.sum(0)
        err = (got - ref).abs().max().item()
        t = timeit(run)
        print(f"  slices {ns:2d}: {t:.1f} us   max |sum(partials) - bf16 result| = {err:.4f}")
