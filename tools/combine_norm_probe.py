"""Graph-timed: moe_combine + add_rmsnorm vs the fused launch at the decode shape (T=32, k=8, H=4096)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
T, k, H = 32, 8, 4096
BF = torch.bfloat16
y = torch.randn(T * k + T, H, device="cuda").to(BF)
pos = torch.randperm(T * k, device="cuda").to(torch.int32)
res = torch.randn(T, H, device="cuda").to(BF); nw = torch.ones(H, device="cuda").to(BF)
def timeit(fn, n=240):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(12): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 12): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 12 * 12) * 1e3
dc = ops.DeferredCombine(y, pos, T, k, T * k, (T, H))
print(f"combine: {timeit(lambda: ops.moe_combine(y, pos, T, k, T * k)):.2f} us")
print(f"combine + add_rmsnorm: {timeit(lambda: ops.add_rmsnorm(ops.moe_combine(y, pos, T, k, T * k), res, nw, 1e-5)):.2f} us")
print(f"fused: {timeit(lambda: ops.moe_combine_add_rmsnorm(dc, res, nw, 1e-5)):.2f} us")
