"""r05: life of every workgroup of the default prefill attention kernel (entry / start-up done / tile loop done / stores done,
s_memrealtime at 100 MHz) -> how long the start and the end of a block are, and how full the 512 workgroup slots are.
Needs: python glm4_moe_nano-vllm_amd/build.py --variant _wgst GLM_FA_WGSTAMP=1 ; GLM4MOE_HIP_LIB=.../libglm4moe_hip_wgst.so"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 4096)
nh, nkv, hd = 96, 8, 128
q = torch.randn(B * S, nh, hd, device="cuda").to(torch.bfloat16)
k = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
v = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
for _ in range(3): ops.varlen_attn_prefill(q, k, v, cu, cu, S, hd ** -0.5)
torch.cuda.synchronize()
lib = ops.load_library()
f = lib.glm_debug_prefill_attn_wgstamps; f.restype = ctypes.c_int; f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
n = min(16384, (S // 128) * nh * B)
buf = (ctypes.c_ulonglong * (16384 * 4))(); assert f(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).view(16384, 4)[:n] / 100.0  # us
t0 = t[:, 0].min(); t = t - t0
span = float(t[:, 3].max())
life = t[:, 3] - t[:, 0]
print(f"{B}x{S}: {n} workgroups, launch span {span:.0f} us; sum of lifetimes / (512 slots x span) = {float(life.sum()) / (512 * span):.3f}")
print(f"  start-up (entry -> first barrier passed): mean {float((t[:,1]-t[:,0]).mean()):.2f} us, p10 {float((t[:,1]-t[:,0]).quantile(0.1)):.2f}, p90 {float((t[:,1]-t[:,0]).quantile(0.9)):.2f}")
print(f"  tile loop: mean {float((t[:,2]-t[:,1]).mean()):.2f} us")
print(f"  output rows (loop end -> stores acknowledged): mean {float((t[:,3]-t[:,2]).mean()):.2f} us, p90 {float((t[:,3]-t[:,2]).quantile(0.9)):.2f}")
nq = S // 128
per_q = nh * B
for lvl in (0, nq // 2, nq - 1):   # launch order: longest first
    sl = slice(lvl * per_q, (lvl + 1) * per_q)
    nt = 2 * (nq - lvl)
    print(f"  query block {nq - 1 - lvl} ({nt} tiles): loop {float((t[sl,2]-t[sl,1]).mean()):.2f} us = {float((t[sl,2]-t[sl,1]).mean()) / nt:.3f} us per tile; start-up {float((t[sl,1]-t[sl,0]).mean()):.2f}; end {float((t[sl,3]-t[sl,2]).mean()):.2f}")
# per-tile cost by a fit over the levels: loop = a + b * ntiles
import numpy as np
x = np.array([2 * (nq - l) for l in range(nq)], dtype=float)
y = np.array([float((t[l * per_q:(l + 1) * per_q, 2] - t[l * per_q:(l + 1) * per_q, 1]).mean()) for l in range(nq)])
b_, a_ = np.polyfit(x, y, 1)
print(f"  fit over query blocks: loop = {a_:.2f} us + {b_:.3f} us x tiles")
# idle slot time: gaps between one workgroup's end and the next workgroup's entry, estimated as span * 512 - sum(life)
print(f"  slot time not inside any workgroup: {100 * (1 - float(life.sum()) / (512 * span)):.1f} % (dispatch gaps + the launch's ragged end)")
