"""fp8 (e4m3) experts at the prefill shape (Air-64E, T = 16384, top-8 + shared): gate|up and down launches, event-timed."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
ops.load_library()
T, E, k, H, I = int(os.environ.get("PROBE_T", 16384)), 64, 8, 4096, 1408
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
w13 = (torch.randn(E + 1, 2 * I, H, device=dev, generator=g, dtype=torch.bfloat16) * 0.02)
w2 = (torch.randn(E + 1, H, I, device=dev, generator=g, dtype=torch.bfloat16) * 0.02)
q13, s13 = ops.quantize_fp8_rowwise(w13)
q2, s2 = ops.quantize_fp8_rowwise(w2)
del w13, w2
x = torch.randn(T, H, device=dev, generator=g, dtype=torch.bfloat16)
idx = torch.stack([torch.randperm(E, device=dev, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
w = (torch.rand(T, k, device=dev, generator=g) + 0.1).to(torch.bfloat16)
offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
R = T * k + T
act = ops.grouped_gemm_gateup_silu_fp8w(x, q13, s13, offs, tok, R)
fns = {"gate|up": (lambda: ops.grouped_gemm_gateup_silu_fp8w(x, q13, s13, offs, tok, R), 2.0 * R * 2 * I * H),
       "down": (lambda: ops.grouped_gemm_down_fp8w(act, q2, s2, offs, scale, E), 2.0 * R * H * I)}
tot = 0.0
for n, (f, fl) in fns.items():
    f(); f(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    t = sorted(ts)[2]
    tot += t
    print(f"T={T} fp8 {n:8s} {t:9.1f} us  {fl / t / 1e6:7.1f} TFLOP/s", flush=True)
print(f"T={T} fp8 pair     {tot:9.1f} us", flush=True)
