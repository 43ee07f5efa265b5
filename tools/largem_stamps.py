"""Phase-segment cycle breakdown of the prefill grouped GEMM (needs the GLM_LM_STAMP=1 variant library:
python glm4_moe_nano-vllm_amd/build.py --variant _stamp GLM_LM_STAMP=1; GLM4MOE_HIP_LIB=.../libglm4moe_hip_stamp.so)."""
import ctypes, os, sys, torch
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
lib = ops.load_library()
f = lib.glm_debug_largem_stamps
f.restype = ctypes.c_int; f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
names = ["dma issue", "lds read issue", "dma wait", "lds wait", "barrier 1", "mfma issue", "barrier 2"]
if os.environ.get("GLM_LARGEM_VARIANT", "").strip() in ("", "-1") or int(os.environ["GLM_LARGEM_VARIANT"]) & 4096:
    # quadrant-phase loop (default): phase 0 in detail, phases 1-3 as one figure
    names = ["p0 read+dma issue", "p0 lds wait", "p0 barrier a", "p0 mfma", "p0 barrier b", "phases 1-3", "-"]
f2 = getattr(lib, "glm_debug_largem_tile_stamps", None)
tnames = ["top-of-tile wait", "K loop", "tile search", "source set-up", "ring fill issue", "epilogue"]
f3 = getattr(lib, "glm_debug_largem_kt_stamps", None)
if f3 is not None:
    f3.restype = ctypes.c_int; f3.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
def show(tag):
    torch.cuda.synchronize()
    if f3 is not None:   # r05: cycles by K-tile position inside an output tile (the K tile that follows a tile switch = 0)
        b26 = (ctypes.c_ulonglong * 26)()
        if f3(b26, 0) == 0 and b26[13]:
            lab = [f"kt {i}" for i in range(6)] + [f"kt nk-{6 - i}" for i in range(6)] + ["others"]
            print(f"{tag} cycles per K tile by position (workgroup 8, wave 0): " +
                  ", ".join(f"{lab[i]} {b26[i] / b26[13 + i]:.0f}" for i in range(13) if b26[13 + i]))
        f3(b26, 1)
    if f2 is not None:
        f2.restype = ctypes.c_int; f2.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
        b8 = (ctypes.c_ulonglong * 8)()
        if f2(b8) == 0 and b8[7]:
            print(f"{tag} tile switch (workgroup 8, wave 0, {b8[7]} tiles), cycles per tile: " +
                  ", ".join(f"{tnames[i]} {b8[i]/b8[7]:.0f}" for i in range(6)))
    buf = (ctypes.c_ulonglong * 16)()
    assert f(buf) == 0
    for g in range(2):
        nk = buf[g * 8 + 7]
        tot = sum(buf[g * 8 + i] for i in range(7))
        print(f"{tag} wave {4*g}: K tiles {nk}, cycles/tile {tot/max(nk,1):.0f}: " +
              ", ".join(f"{names[i]} {buf[g*8+i]/max(nk,1):.0f}" for i in range(7)))
if f3 is not None:
    torch.cuda.synchronize(); f3((ctypes.c_ulonglong * 26)(), 1)
act = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R); show("gate|up")
ops.grouped_gemm_down(act, w2, offs, scale, E); show("down")
if os.environ.get("LM_STAMPS_BF16_ONLY"): sys.exit(0)
# r05: the same for the W8A8 pair on the block-scaled fp8 MFMA
layer.mlp.expert_weight_dtype = "fp8_e4m3"; layer.mlp.quantize_experts_fp8()
w13_q, w13_s, w2_q, w2_s = layer.mlp._fp8
xq, xs = ops.quantize_fp8_rows(x)
r = ops.grouped_gemm_gateup_silu_fp8a8(xq, xs, w13_q, w13_s, offs, tok, R); show("gate|up fp8a8")
ops.grouped_gemm_down_fp8a8(r[0], r[1], w2_q, w2_s, offs, scale, E); show("down fp8a8")
