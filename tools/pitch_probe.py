"""Grouped down projection at the C3 decode shape for several weight row pitches (HBM channel aliasing study)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
BF = torch.bfloat16
T, E, k, H, I = 32, 64, 8, 4096, 1408
g = torch.Generator(device="cuda").manual_seed(3)
idx = torch.stack([torch.randperm(E, device="cuda", generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
wts = (torch.rand(T, k, device="cuda", generator=g) + 0.1).to(BF)
offs, tok, scale, pos = ops.moe_align(idx, wts, E, True)
R = T * k + T
act = torch.randn(R, I, device="cuda").to(BF)
x = torch.randn(T, H, device="cuda").to(BF)
def timeit(fn, n=120):
    fn(0); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for i in range(6): fn(i)
    gph.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 6): gph.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 6 * 6) * 1e3
for name, rows, cols, fn_make in (("down  [G,4096,1408]", H, I, None), ("gate|up [G,2816,4096]", 2 * I, H, None)):
    for pad in (0, 64, 128, 192, 320, 448):
        ws = []
        for i in range(4):
            w = torch.empty(E + 1, rows, cols + pad, dtype=BF, device="cuda").normal_(0, 0.02)[..., :cols]
            ws.append(w)
        if rows == H:
            t = timeit(lambda i: ops.grouped_gemm_down(act, ws[i % 4], offs, scale, E))
        else:
            t = timeit(lambda i: ops.grouped_gemm_gateup_silu(x, ws[i % 4], offs, tok, R))
        print(f"{name} pad {pad:3d} (pitch {(cols + pad) * 2} B): {t:.1f} us")
        del ws
