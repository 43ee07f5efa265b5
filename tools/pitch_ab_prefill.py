"""r05: does the row pitch of the grouped GEMMs' activation-side tensors matter at the prefill shape?  y [R, 4096] and x [T, 4096]
have 8-KiB rows (a multiple of 4 KiB: the HBM channel aliasing that made the weights' padded pitch worth 4.7 -> 5.9 TB/s);
this times down with y at pitch H and H + 64 elements, and gate|up with x at pitch H and H + 64, interleaved."""
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
T, H, E, k = 16384, cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok
x = torch.randn(T, H, device="cuda").to(torch.bfloat16)
idx, w = layer.mlp.gate(x)
offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
w13, w2 = layer.mlp._stacked()
R = T * k + T
I = w2.shape[2]
lib = ops.load_library()
p = ops._p
act = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
def down(ldy):
    store = torch.empty((R, ldy), dtype=torch.bfloat16, device="cuda")
    def f():
        ops._check(lib.glm_grouped_gemm_down(p(act), p(w2), w2.stride(0), w2.stride(1), p(offs), p(scale), E, p(store), ldy, R, H, I,
                                             E + 1, ops._stream()), "down")
    return f, store
def gateup(ldx):
    xs = torch.empty((T, ldx), dtype=torch.bfloat16, device="cuda"); xs[:, :H].copy_(x)
    out = torch.empty((R, I), dtype=torch.bfloat16, device="cuda")
    def f():
        ops._check(lib.glm_grouped_gemm_gateup_silu(p(xs), ldx, p(w13), w13.stride(0), w13.stride(1), p(offs), p(tok), p(out), R, H, I,
                                                    E + 1, ops._stream()), "gateup")
    return f, out
def timeit(f, n=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
d0, y0 = down(H); d1, y1 = down(H + 64)
g0, a0 = gateup(H); g1, a1 = gateup(H + 64)
for r in range(3):
    print(f"round {r}: down y pitch {H}: {timeit(d0):.0f} us | pitch {H + 64}: {timeit(d1):.0f} us || gate|up x pitch {H}: {timeit(g0):.0f} us | pitch {H + 64}: {timeit(g1):.0f} us", flush=True)
assert torch.equal(y0, y1[:, :H]) and torch.equal(a0, a1), "padded pitch changed the results"
print("results identical")
