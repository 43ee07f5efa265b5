"""Where does the fixed cost of a one-token launch go?  (variant library built with GLM_TM_STAMP=1)  Per workgroup of the QKV / gate|up
GEMV: real-time counter (100 MHz) at entry, at its first FMA (first weight unit landed) and at its end."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
lib = ops.load_library()
f = lib.glm_debug_tinym_stamps; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
H, nh, nkv, hd, E, k, I, L = 4096, 96, 8, 128, 64, 8, 1408, 6
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
mk = lambda *s: ops.padded_weight(s[-2], s[-1], *s[:-2], dtype=torch.bfloat16, device=dev).normal_(0.0, 0.02, generator=g)
wqkv = [mk((nh + 2 * nkv) * hd, H) for _ in range(L)]
w13 = [mk(E + 1, 2 * I, H) for _ in range(L)]
x = torch.randn(1, H, device=dev, generator=g).to(torch.bfloat16)
gw = (torch.randn(E, H, device=dev, generator=g) * 0.02).to(torch.bfloat16)
bias = (torch.rand(E, device=dev, generator=g) * 0.2 - 0.1).float()
idx, w = ops.router_topk(x, gw, bias, k, 1.0)
offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
for name, fn in (("qkv", lambda i: ops.gemm_bf16(x, wqkv[i % L], None)),
                 ("gate|up", lambda i: ops.grouped_gemm_gateup_silu(x, w13[i % L], offs, tok, k + 1, max_rows_per_group=1))):
    for i in range(5): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); fn(5); b.record(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (3 * 256))(); assert f(buf, 256) == 0
    st = torch.tensor(list(buf), dtype=torch.float64).view(256, 3) * 0.01  # us
    t0 = float(st[:, 0].min())
    st = st - t0
    print(f"{name}: event time {a.elapsed_time(b) * 1e3:.1f} us; entry {float(st[:,0].min()):.2f}..{float(st[:,0].max()):.2f} us, "
          f"first FMA {float(st[:,1].min()):.2f}..{float(st[:,1].max()):.2f} (median {float(st[:,1].median()):.2f}), "
          f"end {float(st[:,2].min()):.2f}..{float(st[:,2].max()):.2f} (median {float(st[:,2].median()):.2f})")
    e = st[:, 2]
    print("  end by XCD (blockIdx % 8):   " + " ".join(f"{float(e[x::8].mean()):.1f}" for x in range(8)))
    print("  end by blockIdx / 32:        " + " ".join(f"{float(e[32 * j:32 * j + 32].mean()):.1f}" for j in range(8)))
    print("  first FMA by blockIdx / 32:  " + " ".join(f"{float(st[32 * j:32 * j + 32, 1].mean()):.1f}" for j in range(8)))
    print("  streaming time (end - first FMA) by blockIdx / 32: " + " ".join(f"{float((st[32 * j:32 * j + 32, 2] - st[32 * j:32 * j + 32, 1]).mean()):.1f}" for j in range(8)))
    print("  end, every 8th workgroup:    " + " ".join(f"{float(v):.1f}" for v in e[::8]))
