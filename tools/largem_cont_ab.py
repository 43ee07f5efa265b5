"""r05: same-process A/B of the grouped prefill GEMMs (T = 16384, Air dims) with and without LM_CONT (the next output tile's
first K tiles issued from the current tile's last two): option largem_variant 86787 | 611075, interleaved rounds."""
import os, sys, torch
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
layer = model.model.layers[1]
variants = [int(v) for v in sys.argv[1:]] or [86787, 611075]
for rnd in range(3):
    for v in variants:
        with hip_ops.option("largem_variant", v):
            r = bench.measure_prefill_gemm(layer, cfg, hip_ops, 1, torch.device("cuda"), iters=6)
        print(f"round {rnd} variant {v}: " + "  ".join(f"{k.split('@')[0]} {x['kernel_us_mean']:.0f} us {x['achieved']:.0f} TFLOP/s" for k, x in r.items()), flush=True)
