"""Decode-shape grouped expert GEMMs: bf16 vs fp8 weight streaming (same box, same routing)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import bench
from nanovllm import hip_ops as ops
from nanovllm.models.glm4_moe.model import Glm4MoeForCausalLM
torch.set_default_dtype(torch.bfloat16)
L = 6
cfg = bench.air64e_config(L, 4096)
cfg.keep_bf16_experts = True  # this tool times the bf16 and the fp8 stacks side by side
with torch.device("cuda"):
    model = Glm4MoeForCausalLM(cfg)
bench.init_weights(model, 1)
layers = [l for l in model.model.layers if hasattr(l.mlp, "gate")]
B, E, k = 32, 64, 8
x = torch.randn(B, 4096, device="cuda").to(torch.bfloat16)
prep = []
for l in layers:
    idx, w = l.mlp.gate(x)
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    l.mlp.quantize_experts_fp8()
    prep.append((offs, tok, scale, pos) + tuple(l.mlp._stacked()) + l.mlp._fp8)
R = B * k + B
def timeit(fn, n=12):
    fn(0); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
acts = {}
def gu16(i):
    p = prep[i % len(prep)]; acts[i % len(prep)] = ops.grouped_gemm_gateup_silu(x, p[4], p[0], p[1], R)
def gu8(i):
    p = prep[i % len(prep)]; acts[i % len(prep)] = ops.grouped_gemm_gateup_silu_fp8w(x, p[6], p[7], p[0], p[1], R)
def dn16(i):
    p = prep[i % len(prep)]; ops.grouped_gemm_down(acts[i % len(prep)], p[5], p[0], p[2], E)
def dn8(i):
    p = prep[i % len(prep)]; ops.grouped_gemm_down_fp8w(acts[i % len(prep)], p[8], p[9], p[0], p[2], E)
for name, f in (("gate|up bf16", gu16), ("gate|up fp8w", gu8), ("down bf16", dn16), ("down fp8w", dn8)):
    print(f"{name:14s} {timeit(f):8.1f} us")
