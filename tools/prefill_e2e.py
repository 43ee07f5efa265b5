"""Prefill throughput of the layer path: T tokens (4 x 4096) through L decoder layers (eager)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import bench
from nanovllm.models.glm4_moe.model import Glm4MoeForCausalLM
from nanovllm.utils.context import set_context
L, B, S = 6, 4, 4096
torch.set_default_dtype(torch.bfloat16)
cfg = bench.air64e_config(L, 4096)
with torch.device("cuda"):
    model = Glm4MoeForCausalLM(cfg)
bench.init_weights(model, 1)
T = B * S
nblk = B * 16
kv = torch.zeros(2, L, nblk, 256, 8, 128, device="cuda")
lid = 0
for m in model.modules():
    if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
        m.k_cache, m.v_cache = kv[0, lid], kv[1, lid]; lid += 1
ids = torch.randint(0, cfg.vocab_size, (T,), device="cuda")
pos = torch.arange(S, device="cuda").repeat(B)
cu = torch.arange(0, T + 1, S, dtype=torch.int32, device="cuda")
slots = torch.arange(T, dtype=torch.int32, device="cuda")
set_context(True, cu, cu, S, S, slots, None, None)
@torch.inference_mode()
def step():
    return model.compute_logits(model(ids, pos))
step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"prefill {T} tokens x {L} layers: {dt*1e3:.1f} ms/step = {dt*1e3/L:.2f} ms/layer -> {T/dt:.0f} tok/s at L={L}; "
      f"extrapolated L=46: {T/(dt/L*46):.0f} tok/s")
