"""Prefill-shape grouped gate|up GEMM: TFLOP/s of successive groups of 5 launches (boost vs sustained clocks)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import bench
from nanovllm import hip_ops as ops
from nanovllm.models.glm4_moe.model import Glm4MoeForCausalLM
torch.set_default_dtype(torch.bfloat16)
cfg = bench.air64e_config(2, 4096)
with torch.device("cuda"):
    model = Glm4MoeForCausalLM(cfg)
bench.init_weights(model, 1)
layer = model.model.layers[1]
T, H, E, k, I = 16384, cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
x = torch.randn(T, H, device="cuda").to(torch.bfloat16)
idx, w = layer.mlp.gate(x)
offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
w13, w2 = layer.mlp._stacked()
R = T * k + T
flops = 2.0 * R * 2 * I * H
ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R); torch.cuda.synchronize()
out = []
for grp in range(12):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
    b.record(); torch.cuda.synchronize()
    out.append(flops / (a.elapsed_time(b) / 5 * 1e-3) / 1e12)
print("gate|up TFLOP/s per group of 5 launches:", " ".join(f"{v:.0f}" for v in out))
