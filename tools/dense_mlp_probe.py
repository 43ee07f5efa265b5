"""Dense-MLP (layer 0) projections at decode: gate|up [2*10944, 4096] and down [4096, 10944] with one group of 32 rows."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
BF = torch.bfloat16
T, H, I = 32, 4096, 10944
x = torch.randn(T, H, device="cuda").to(BF)
def timeit(fn, n=120):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(6): fn(i)
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 6): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 6 * 6) * 1e3
w13 = [ops.padded_weight(2 * I, H, 1, dtype=BF, device="cuda").normal_(0, 0.02) for _ in range(3)]
w2 = [ops.padded_weight(H, I, 1, dtype=BF, device="cuda").normal_(0, 0.02) for _ in range(3)]
offs = torch.tensor([0, T], dtype=torch.int32, device="cuda")
act = ops.grouped_gemm_gateup_silu(x, w13[0], offs, None, T)
t1 = timeit(lambda i: ops.grouped_gemm_gateup_silu(x, w13[i % 3], offs, None, T))
t2 = timeit(lambda i: ops.grouped_gemm_down(act, w2[i % 3], offs, None, 0))
print(f"dense gate|up: {t1:.1f} us ({2*I*H*2/t1/1e3:.0f} GB/s)   dense down: {t2:.1f} us ({H*I*2/t2/1e3:.0f} GB/s)")
