"""Prefill QKV projection + split/rotary/cache store: one launch (glm_gemm_qkv_rope_store) vs two, T = 16384, Air dims."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
T, H, nh, nkv, hd, rd = 16384, 4096, 96, 8, 128, 64
N = (nh + 2 * nkv) * hd
x = torch.randn(T, H, device="cuda").to(torch.bfloat16)
w = ops.padded_weight(N, H, dtype=torch.bfloat16, device="cuda"); w.normal_(0, 0.02)
b = (torch.randn(N, device="cuda") * 0.02).to(torch.bfloat16)
pos = torch.arange(4096, device="cuda").repeat(4)
cs = torch.randn(4096, rd, device="cuda")
kc = torch.zeros(72, 256, nkv, hd, device="cuda", dtype=torch.bfloat16); vc = torch.zeros_like(kc)
slots = torch.arange(T, dtype=torch.int32, device="cuda")
def two():
    qkv = ops.gemm_bf16(x, w, b)
    return ops.qkv_split_norm_rope_store(qkv, nh, nkv, hd, pos, cs, rd, None, None, 1e-5, kc, vc, slots, True)
def one():
    return ops.gemm_qkv_rope_store(x, w, b, nh, nkv, hd, pos, cs, rd, kc, vc, slots, True)
def t(f, n=10, warm=4):
    for _ in range(warm): f()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
for r in range(3):
    print(f"round {r}: two launches {t(two):.0f} us, one launch {t(one):.0f} us", flush=True)
