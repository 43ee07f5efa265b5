"""Decode attention time vs kv-split count at small batch (graph-timed, so host launch cost is excluded)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
BF = torch.bfloat16
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
nh, nkv, hd, bs = 96, 8, 128, 256
for B, ctx in ((1, 1024), (1, 4096), (4, 1024), (4, 4096), (8, 4096)):
    blocks = (ctx + bs - 1) // bs
    kc = torch.randn(B * blocks + 2, bs, nkv, hd, device="cuda").to(BF); vc = torch.randn_like(kc)
    bt = torch.arange(B * blocks, dtype=torch.int32, device="cuda").view(B, blocks)
    cl = torch.full((B,), ctx, dtype=torch.int32, device="cuda")
    q = torch.randn(B, nh, hd, device="cuda").to(BF)
    res = []
    for s in (0, 1, 2, 4, 8, 16, 32):
        if s > ctx // 32: continue
        t = timeit(lambda: ops.paged_attn_decode(q, kc, vc, cl, bt, hd ** -0.5, s, ctx))
        res.append(f"{'auto' if s == 0 else s}: {t:.1f}")
    print(f"B={B} ctx={ctx}: " + "  ".join(res) + "  (us)")
