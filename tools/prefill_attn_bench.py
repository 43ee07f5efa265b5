"""Time the varlen causal prefill attention at the C3 prefill shape (4 x 4096 tokens, 96/8 heads)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 4096)
nh, nkv, hd = 96, 8, 128
q = torch.randn(B * S, nh, hd, device="cuda").to(torch.bfloat16)
k = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
v = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
f = lambda: ops.varlen_attn_prefill(q, k, v, cu, cu, S, hd ** -0.5)
for _ in range(4): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(12): f()
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 12
flops = 4 * B * nh * S * S * hd / 2  # causal: half the score matrix
print(f"prefill attn {B}x{S}: {ms*1e3:.0f} us, {flops/ms/1e9:.1f} TFLOP/s ({flops/ms/1e9/2500:.1%} of bf16 peak)")
