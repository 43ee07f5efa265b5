"""Single-sequence decode (T = 1, 2): the GEMV body (csrc/gemm_tinym.hip) against the small-M MFMA kernels the same calls
take with GLM4MOE_TINYM=0 — QKV, o_proj, grouped gate|up and down at Air dims, weights rotated over 6 layers' worth so
that every launch streams cold bytes.  One process, interleaved rounds."""
import os, sys, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
ops.load_library()
H, nh, nkv, hd, E, k, I, L = 4096, 96, 8, 128, 64, 8, 1408, int(os.environ.get("PROBE_L", "6"))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
mk = lambda *s: ops.padded_weight(s[-2], s[-1], *s[:-2], dtype=torch.bfloat16, device=dev).normal_(0.0, 0.02, generator=g)
wqkv = [mk((nh + 2 * nkv) * hd, H) for _ in range(L)]
bq = torch.randn((nh + 2 * nkv) * hd, device=dev, generator=g).to(torch.bfloat16)
wo = [mk(H, nh * hd) for _ in range(L)]
w13 = [mk(E + 1, 2 * I, H) for _ in range(L)]
w2 = [mk(E + 1, H, I) for _ in range(L)]
gw = (torch.randn(E, H, device=dev, generator=g) * 0.02).to(torch.bfloat16)
bias = (torch.rand(E, device=dev, generator=g) * 0.2 - 0.1).float()
out = {}
for T in (1, 2):
    x = torch.randn(T, H, device=dev, generator=g).to(torch.bfloat16)
    xo = torch.randn(T, nh * hd, device=dev, generator=g).to(torch.bfloat16)
    idx, w = ops.router_topk(x, gw, bias, k, 1.0)
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu(x, w13[0], offs, tok, R, max_rows_per_group=T)
    fns = {
        "qkv": (lambda i: ops.gemm_bf16(x, wqkv[i % L], bq), (nh + 2 * nkv) * hd * H * 2),
        "o_proj": (lambda i: ops.gemm_bf16(xo, wo[i % L], None), H * nh * hd * 2),
        "gateup": (lambda i: ops.grouped_gemm_gateup_silu(x, w13[i % L], offs, tok, R, max_rows_per_group=T), None),
        "down": (lambda i: ops.grouped_gemm_down(act, w2[i % L], offs, scale, E, max_rows_per_group=T), None),
    }
    touched = int(((offs[1:E + 2] - offs[:E + 1]) > 0).sum())
    nbytes = {"qkv": fns["qkv"][1], "o_proj": fns["o_proj"][1], "gateup": touched * 2 * I * H * 2, "down": touched * H * I * 2}
    res = {n: {"tinym": [], "smallm": []} for n in fns}
    for rnd in range(4):
        for mode in ("tinym", "smallm"):
            os.environ["GLM4MOE_TINYM"] = "1" if mode == "tinym" else "0"
            for n, (f, _) in fns.items():
                f(0); f(1)
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for i in range(12): f(i)
                b.record(); torch.cuda.synchronize()
                res[n][mode].append(a.elapsed_time(b) / 12 * 1e3)
    os.environ["GLM4MOE_TINYM"] = "1"
    for n in fns:
        t, s = sorted(res[n]["tinym"])[1], sorted(res[n]["smallm"])[1]
        out[f"T{T}.{n}"] = {"tinym_us": round(t, 2), "smallm_us": round(s, 2), "MB": round(nbytes[n] / 1e6, 1),
                            "tinym_TBps": round(nbytes[n] / t / 1e6, 2), "smallm_TBps": round(nbytes[n] / s / 1e6, 2)}
        print(f"T={T} {n:8s} tinym {t:7.2f} us ({nbytes[n]/t/1e6:.2f} TB/s)   small-M {s:7.2f} us ({nbytes[n]/s/1e6:.2f} TB/s)", flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tinym_probe.json"), "w"), indent=1)
