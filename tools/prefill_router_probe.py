import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import torch
from nanovllm import hip_ops as ops
ops.load_library()
T, H, E, k = 16384, 4096, 64, 8
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(T, H, device="cuda", generator=g).to(torch.bfloat16)
res = torch.randn(T, H, device="cuda", generator=g).to(torch.bfloat16)
nw = torch.ones(H, device="cuda", dtype=torch.bfloat16)
gw = (torch.randn(E, H, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
bias = (torch.rand(E, device="cuda", generator=g) * 0.2 - 0.1).float()
def timed(f, n=10):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("fused norm+router+align", timed(lambda: ops.add_rmsnorm_router_topk(x, res, nw, 1e-5, gw, bias, k, 1.0, align_with_shared=True)))
print("add_rmsnorm", timed(lambda: ops.add_rmsnorm(x, res, nw, 1e-5)))
y, ro = ops.add_rmsnorm(x, res, nw, 1e-5)
print("gemm logits", timed(lambda: ops.gemm_bf16(y, gw, None)))
lg = ops.gemm_bf16(y, gw, None)
ref = (y.float() @ gw.float().t()).to(torch.bfloat16)
print("logits max ulp-ish diff", float((lg.float() - ref.float()).abs().max()), "frac differing", float((lg != ref).float().mean()))
print("router standalone", timed(lambda: ops.router_topk(y, gw, bias, k, 1.0)))
idx, w = ops.router_topk(y, gw, bias, k, 1.0)
print("moe_align", timed(lambda: ops.moe_align(idx, w, E, True)))
