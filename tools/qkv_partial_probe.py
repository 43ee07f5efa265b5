"""Would QKV pay as K-slice partial sums (reduced by its consumer, as o_proj's are)?  Times glm_gemm_bf16_partial at the QKV
shape (N=14336, K=4096) for 1/2/4/8 slices against the one-pass glm_gemm_bf16, M=32 and M=1, rotating over enough weight
copies that every launch streams cold data.    python tools/qkv_partial_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import torch

from nanovllm import hip_ops as ops


def timed(fns, reps=6):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        for f in fns:
            f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (reps * len(fns)) * 1e3


def main():
    ops.load_library()
    N, K, copies = 14336, 4096, 24  # 24 x 117 MB = 2.8 GB >> L2 + MALL
    ws = [ops.padded_weight(N, K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    bias = torch.zeros(N, dtype=torch.bfloat16, device="cuda")
    for M in (32, 1):
        x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
        t0 = timed([lambda w=w: ops.gemm_bf16(x, w, bias) for w in ws])
        line = f"M={M:2d}: one pass {t0:6.1f} us"
        for want in (2, 4, 8):
            t = timed([lambda w=w: ops.gemm_bf16_partial(x, w, want) for w in ws])
            line += f" | {want} slices {t:6.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
