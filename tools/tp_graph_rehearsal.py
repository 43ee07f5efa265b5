"""Single-GPU rehearsal of the TP decode step under HIP-graph capture WITH torch.distributed (RCCL) collectives.

A one-rank "nccl" process group is created and the TP coordinator is forced to issue its all-reduce / gather even
though world_size == 1, so that the capture path the N>1 bench takes (ProcessGroupNCCL work objects, watchdog thread,
RCCL kernels recorded into the graph) is exercised on the one GPU this box has.  Prints eager-vs-graph token equality
and the CPU time needed to enqueue one eager step (the launch-bound floor of an eager TP step)."""
import os, sys, time, torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

import bench
from nanovllm import hip_ops
from nanovllm.distributed import parallel_state
from nanovllm.models.glm4_moe.model import Glm4MoeForCausalLM
from nanovllm.utils.context import set_context

calls = {"n": 0}
def forced_all_reduce(self, t):
    calls["n"] += 1
    dist.all_reduce(t)
    return t
parallel_state.TPGroup.all_reduce = forced_all_reduce

L, B, ctx, bs = 6, 32, 1024, 256
cfg = bench.air64e_config(L, 4096)
torch.set_default_dtype(torch.bfloat16)
with torch.device(dev):
    model = Glm4MoeForCausalLM(cfg)
bench.init_weights(model, 1)
for m in model.modules():                      # make the row-parallel sites believe they are sharded
    if hasattr(m, "tp_size") and type(m).__name__ in ("RowParallelLinear",):
        m.tp_size = 2
blocks = (ctx + bs - 1) // bs
nblk = B * blocks + 4
kv = torch.randn(2, L, nblk, bs, cfg.num_key_value_heads, cfg.head_dim, device=dev)
lid = 0
for m in model.modules():
    if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
        m.k_cache, m.v_cache = kv[0, lid], kv[1, lid]; lid += 1
bt = torch.arange(B * blocks, dtype=torch.int32, device=dev).view(B, blocks)
cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
pos = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)
slot = (bt[:, (ctx - 1) // bs].long() * bs + (ctx - 1) % bs).to(torch.int32)
ids = torch.randint(0, cfg.vocab_size, (B,), device=dev)
set_context(False, slot_mapping=slot, context_lens=cl, block_tables=bt)
out = torch.zeros(B, dtype=torch.int64, device=dev)

@torch.inference_mode()
def step():
    h = model(ids, pos)
    logits = model.compute_logits(h)
    out.copy_(torch.argmax(logits, -1))

step(); torch.cuda.synchronize()
print("all-reduce calls per step:", calls["n"])
eager = out.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
out.zero_(); g.replay(); torch.cuda.synchronize()
print("graph == eager tokens:", bool(torch.equal(out, eager)))
for name, fn in (("eager", step), ("graph", g.replay)):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: enqueue {1e3*(t1-t0)/20:.2f} ms/step, total {1e3*(t2-t0)/20:.2f} ms/step ({L} layers)")
dist.destroy_process_group()
