"""Is the large-M loop waiting for memory?  The grouped gate|up / down launches at T = 16384 with the SAME tile shapes and K but
fewer, larger experts: (E, k) = (64, 8) is the product shape (1.5 GB of expert weights per launch, every W panel read by 8-9 row
blocks), (8, 8) and (2, 2) send every token to every expert (207 / 69 MB of weights: resident in the 256 MB memory-side cache, each
W panel re-read by 64 row blocks).  Same MFMA work per row; only where the operands come from changes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
ops.load_library()
T, H, I = 16384, 4096, 1408
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(T, H, device=dev, generator=g, dtype=torch.bfloat16)
for E, k in ((64, 8), (8, 8), (2, 2)):
    w13 = ops.padded_weight(2 * I, H, E + 1, dtype=torch.bfloat16, device=dev).normal_(0.0, 0.02, generator=g)
    w2 = ops.padded_weight(H, I, E + 1, dtype=torch.bfloat16, device=dev).normal_(0.0, 0.02, generator=g)
    idx = torch.stack([torch.randperm(E, device=dev, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, device=dev, generator=g) + 0.1).to(torch.bfloat16)
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
    for n, f, fl in (("gate|up", lambda: ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R), 2.0 * R * 2 * I * H),
                     ("down", lambda: ops.grouped_gemm_down(act, w2, offs, scale, E), 2.0 * R * H * I)):
        for _ in range(4): f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(6): f()
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / 6 * 1e3
        print(f"E={E:3d} k={k} R={R:7d} {n:8s} {t:9.1f} us  {fl / t / 1e6:7.1f} TFLOP/s", flush=True)
    del w13, w2, act
