"""Time add_rmsnorm against add_rmsnorm_partial (4 and 8 slices) and the o_proj + norm pairs at the decode shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
T, H, K = 32, 4096, 12288
BF = torch.bfloat16
x = torch.randn(T, K, device="cuda").to(BF)
ws = []
for i in range(6):
    w = ops.padded_weight(H, K, dtype=BF, device="cuda"); w.copy_(torch.randn(H, K, device="cuda") * K ** -0.5); ws.append(w)
res = torch.randn(T, H, device="cuda").to(BF); nw = torch.ones(H, device="cuda").to(BF)
def timeit(fn, n=200):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(12): fn(i)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 12): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 12 * 12) * 1e3
y = ops.gemm_bf16(x, ws[0], None)
print(f"add_rmsnorm alone: {timeit(lambda i: ops.add_rmsnorm(y, res, nw, 1e-5)):.2f} us")
for ns in (2, 4, 8):
    ps = ops.gemm_bf16_partial(x, ws[0], ns)
    print(f"add_rmsnorm_partial({ps.part.shape[0]}) alone: {timeit(lambda i: ops.add_rmsnorm_partial(ps, res, nw, 1e-5)):.2f} us")
print(f"pair gemm_bf16 + add_rmsnorm: {timeit(lambda i: ops.add_rmsnorm(ops.gemm_bf16(x, ws[i % 6], None), res, nw, 1e-5)):.2f} us")
for ns in (3, 4, 6, 8):
    print(f"pair partial({ns}) + fused norm: {timeit(lambda i: ops.add_rmsnorm_partial(ops.gemm_bf16_partial(x, ws[i % 6], ns), res, nw, 1e-5)):.2f} us")
