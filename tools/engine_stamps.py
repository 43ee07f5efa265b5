"""r05: where the one-launch MLP half of the one-token decode layer (csrc/decode_engine.hip) spends its time.
Build:  python glm4_moe_nano-vllm_amd/build.py --variant _enstamp GLM_EN_STAMP=1
Run:    GLM4MOE_HIP_LIB=$PWD/glm4_moe_nano-vllm_amd/lib/libglm4moe_hip_enstamp.so python tools/engine_stamps.py
Prints, per phase edge, the median / max over the 256 workgroups of the 100 MHz real-time stamps (us from the earliest
workgroup's entry), and event timings of the engine launch against the three launches it replaces, rotating over several
weight sets so that every launch streams cold weights."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
BF = torch.bfloat16
K, H, E, k, I = 12288, 4096, 128, 8, 1408
NSET = 4
g = torch.Generator(device="cuda").manual_seed(1)
def rn(*s, sc=1.0): return (torch.randn(*s, device="cuda", generator=g) * sc).to(BF)
sets = []
for i in range(NSET):
    wo = ops.padded_weight(H, K, dtype=BF, device="cuda"); wo.normal_(0, K ** -0.5, generator=g)
    w13 = ops.padded_weight(2 * I, H, E + 1, dtype=BF, device="cuda"); w13.normal_(0, H ** -0.5, generator=g)
    w2 = ops.padded_weight(H, I, E + 1, dtype=BF, device="cuda"); w2.normal_(0, I ** -0.5, generator=g)
    sets.append((wo, rn(H) * 0.1 + 1.0, rn(E, H, sc=H ** -0.5), (torch.rand(E, device="cuda", generator=g) * 0.2 - 0.1).float(), w13, w2))
x, res = rn(1, K), rn(1, H)
def engine(i):
    wo, nw, gw, b, w13, w2 = sets[i % NSET]
    return ops.decode_mlp_engine(x, wo, res, nw, 1e-5, gw, b, k, 2.5, w13, w2)
def three(i):
    wo, nw, gw, b, w13, w2 = sets[i % NSET]
    yn, ro, idx, tw, (offs, tok, scale, pos) = ops.gemm_add_rmsnorm_router_topk_align(x, wo, res, nw, 1e-5, gw, b, k, 2.5, True)
    act = ops.grouped_gemm_gateup_silu(yn.view(1, H), w13, offs, tok, k + 1, max_rows_per_group=1)
    return ops.grouped_gemm_down(act, w2, offs, scale, E, max_rows_per_group=1)
def timeit(fn, n=20):
    fn(0); torch.cuda.synchronize()
    best = []
    for rep in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(n): fn(i)
        b.record(); torch.cuda.synchronize()
        best.append(a.elapsed_time(b) / n * 1e3)
    return min(best), sum(best) / len(best)
print("three launches  us/call (best, mean):", "%.1f %.1f" % timeit(three))
print("engine          us/call (best, mean):", "%.1f %.1f" % timeit(engine))
ops.check_decode_engine()
lib = ops.load_library()
if hasattr(lib, "glm_debug_engine_stamps"):
    names = ["entry", "o_proj done (wave 0)", "seam A gathered", "norm done", "routing ready (wave 7)", "act gathered (wave 7)", "end"]
    for rep in range(3):
        engine(rep); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (256 * 8))()
        assert lib.glm_debug_engine_stamps(buf, 256) == 0
        t = torch.tensor(list(buf), dtype=torch.float64).view(256, 8)[:, :7] / 100.0  # us
        t0 = t[:, 0].min()
        t = t - t0
        print(f"--- launch {rep}: us since the first workgroup's entry (median / min / max over 256 workgroups)")
        for j, nme in enumerate(names):
            c = t[:, j]
            print(f"  {nme:28s} {c.median():7.2f} {c.min():7.2f} {c.max():7.2f}")
