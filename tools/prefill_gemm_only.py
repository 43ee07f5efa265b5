"""Run only the prefill-shape grouped GEMMs a few times (for rocprofv3 --pmc passes)."""
import os, sys, types, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import bench
from nanovllm import hip_ops
from nanovllm.models.glm4_moe.model import Glm4MoeForCausalLM
torch.set_default_dtype(torch.bfloat16)
cfg = bench.air64e_config(2, 4096)
with torch.device("cuda"):
    model = Glm4MoeForCausalLM(cfg)
bench.init_weights(model, 1)
mode = os.environ.get("PG_MODE", "bf16")   # bf16 | fp8w (weight-only) | a8 (W8A8 on the block-scaled fp8 MFMA, r05)
layer = model.model.layers[1]
if mode != "bf16":
    layer.mlp.expert_weight_dtype = "fp8_e4m3"
    layer.mlp.quantize_experts_fp8()
r = bench.measure_prefill_gemm(layer, cfg, hip_ops, 1, torch.device("cuda"), iters=2, fp8=mode == "fp8w", a8=mode == "a8")
for k, v in r.items():
    print(k, v["achieved"], v["kernel_us_mean"])
