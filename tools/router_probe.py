"""Graph-timed router / fused norm+router at the decode shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
T, H, E, k = 32, 4096, 64, 8
BF = torch.bfloat16
x = torch.randn(T, H, device="cuda").to(BF); res = torch.randn(T, H, device="cuda").to(BF)
nw = torch.ones(H, device="cuda").to(BF); gw = (torch.randn(E, H, device="cuda") * H ** -0.5).to(BF)
bias = torch.zeros(E, device="cuda")
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
print(f"router_topk: {timeit(lambda: ops.router_topk(x, gw, bias, k, 1.0)):.2f} us")
print(f"add_rmsnorm + router_topk: {timeit(lambda: ops.router_topk(ops.add_rmsnorm(x, res, nw, 1e-5)[0], gw, bias, k, 1.0)):.2f} us")
print(f"fused: {timeit(lambda: ops.add_rmsnorm_router_topk(x, res, nw, 1e-5, gw, bias, k, 1.0)):.2f} us")
