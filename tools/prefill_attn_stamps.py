"""Per-segment cycle breakdown of the prefill attention loop (GLM_FA_STAMP=1 variant library)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
B, S, nh, nkv, hd = 4, 4096, 96, 8, 128
q = torch.randn(B * S, nh, hd, device="cuda").to(torch.bfloat16)
k = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
v = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
for _ in range(2): ops.varlen_attn_prefill(q, k, v, cu, cu, S, hd ** -0.5)
torch.cuda.synchronize()
lib = ops.load_library()
f = lib.glm_debug_prefill_attn_stamps; f.restype = ctypes.c_int; f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * 16)(); assert f(buf) == 0
names = ["S issue", "LDS read+DMA issue", "softmax", "V wait", "PV issue", "K wait", "S issue+DMA", "softmax+PV",
         "S issue+V read", "V wait", "DMA wait", "barrier", "DMA+K read issue", "softmax+PV", "K wait+S+DMA", "softmax+PV"]
nt = S // 64
tot = sum(buf)
print(f"workgroup 0 wave 0: {nt} tiles, {tot/nt:.0f} cycles per 64-key tile (4 units)")
for i, n in enumerate(names): print(f"  {i:2d} {n:22s} {buf[i]/nt:7.0f}")
