"""r05: the W8A8 expert GEMM pair on the block-scaled fp8 MFMA at Air dims, T = 16384 (one process): time of gate|up, down
and the activation quantisation launch, against the weight-only fp8 pair and the bf16 pair.  python tools/fp8a8_probe.py"""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import bench
from nanovllm import hip_ops as ops
from nanovllm.models.glm4_moe.decode_layer import Glm4MoeDecoderLayer
torch.set_default_dtype(torch.bfloat16)
cfg = bench.air64e_config(2, 4096)
dev = torch.device("cuda", 0)
with torch.device(dev):
    layer = Glm4MoeDecoderLayer(cfg, prefix="model.layers.1")
g = torch.Generator(device="cuda").manual_seed(3)
for n, p in layer.named_parameters():
    p.data.normal_(0.0, 0.02, generator=g)
T = int(os.environ.get("T", "16384"))
r = {"bf16": bench.measure_prefill_gemm(layer, cfg, ops, 1, dev, T=T, iters=6, warm=3)}
layer.mlp.expert_weight_dtype = "fp8_e4m3"; layer.mlp.quantize_experts_fp8()
r["fp8_weight_only"] = bench.measure_prefill_gemm(layer, cfg, ops, 1, dev, T=T, iters=6, warm=3, fp8=True)
r["fp8_w8a8"] = bench.measure_prefill_gemm(layer, cfg, ops, 1, dev, T=T, iters=6, warm=3, a8=True)
for k, v in r.items():
    print(k)
    for kk, vv in v.items():
        print(f"   {kk:40s} {vv['kernel_us_mean']:9.1f} us {vv['achieved']:8.1f} TFLOP/s  frac {vv['frac']:.4f} of {vv['peak']:.0f}"
              + (f"   (+ quantize x {vv['quantize_x_us']} us -> {vv['achieved_incl_quantize_x']} TFLOP/s)" if "quantize_x_us" in vv else ""))
