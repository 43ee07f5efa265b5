"""A/B of the gemm_largem kernel variants at the prefill shape of BASELINE config 3 (T=16384 -> 147456 rows, Air dims)
in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): glm_debug_largem_variant() switches the variant
between launches.  Every variant that stores its results is compared bit for bit with variant 0 first.

    python tools/largem_ab.py [--variants 0,1,2,3,4,7] [--rounds 4] [--iters 5] [--out gpurun_out/largem_ab.json]
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
# the variant / band switches exist only in the study build of the library:
#     python glm4_moe_nano-vllm_amd/build.py --variant _study GLM_LM_STUDY=1
_STUDY = os.path.join(ROOT, "glm4_moe_nano-vllm_amd", "lib", "libglm4moe_hip_study.so")
os.environ.setdefault("GLM4MOE_HIP_LIB", _STUDY)
if not os.path.exists(os.environ["GLM4MOE_HIP_LIB"]):
    raise SystemExit("build the study library first: python glm4_moe_nano-vllm_amd/build.py --variant _study GLM_LM_STUDY=1")
import torch

from nanovllm import hip_ops as ops

NOSTORE = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,3,4099")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tokens", type=int, default=16384)
    ap.add_argument("--band", type=int, default=0, help="row blocks per L2 band (0: the library default)")
    ap.add_argument("--stress", type=int, default=0,
                    help="race screen: this many extra launches of every storing variant, each compared bit for bit with variant 0")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "largem_ab.json"))
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    lib = ops.load_library()
    setvar = lib.glm_debug_largem_variant
    setvar.restype, setvar.argtypes = ctypes.c_int, [ctypes.c_int]
    lib.glm_debug_largem_band.restype, lib.glm_debug_largem_band.argtypes = ctypes.c_int, [ctypes.c_int]
    if args.band > 0:
        lib.glm_debug_largem_band(args.band)

    H, E, k, I, T = 4096, 64, 8, 1408, args.tokens
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(T, H, device="cuda", generator=g).to(torch.bfloat16)
    gw = (torch.randn(E, H, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    bias = (torch.rand(E, device="cuda", generator=g) * 0.2 - 0.1).float()
    w13 = ops.padded_weight(2 * I, H, E + 1, dtype=torch.bfloat16, device="cuda")
    w2 = ops.padded_weight(H, I, E + 1, dtype=torch.bfloat16, device="cuda")
    w13.normal_(0.0, 0.02, generator=g)
    w2.normal_(0.0, 0.02, generator=g)
    idx, w = ops.router_topk(x, gw, bias, k, 1.0)
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    R = T * k + T
    fl = {"gateup": 2.0 * R * 2 * I * H, "down": 2.0 * R * H * I}

    setvar(0)
    act0 = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
    y0 = ops.grouped_gemm_down(act0, w2, offs, scale, E)
    torch.cuda.synchronize()
    fns = {"gateup": lambda: ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R),
           "down": lambda: ops.grouped_gemm_down(act0, w2, offs, scale, E)}
    parity = {}
    for v in variants:
        if v & NOSTORE:
            continue
        setvar(v)
        a = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
        y = ops.grouped_gemm_down(act0, w2, offs, scale, E)
        torch.cuda.synchronize()
        parity[v] = bool(torch.equal(a, act0)) and bool(torch.equal(y, y0))
        print(f"variant {v}: bit-identical to variant 0: {parity[v]}", flush=True)
    stress = {}
    if args.stress > 0:
        # a sync-structure edit must be screened for races over many runs (cdna_hip_programming.md, 8-phase template):
        # a DMA that lands late or a slot refilled early shows as a rare wrong tile, so every launch is compared in full
        for v in variants:
            if v & NOSTORE or v == 0:
                continue
            setvar(v)
            bad = 0
            for i in range(args.stress):
                a = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
                y = ops.grouped_gemm_down(act0, w2, offs, scale, E)
                if not (torch.equal(a, act0) and torch.equal(y, y0)):
                    bad += 1
            stress[v] = {"launches": args.stress, "mismatching_launches": bad}
            print(f"variant {v}: {args.stress} stress launches, {bad} differ from variant 0", flush=True)
    # warm the clocks on MFMA-bound work, then interleaved rounds
    setvar(0)
    for _ in range(5):
        fns["gateup"]()
        fns["down"]()
    torch.cuda.synchronize()
    times = {name: {v: [] for v in variants} for name in fns}
    for r in range(args.rounds):
        for name, fn in fns.items():
            for v in variants:
                setvar(v)
                fn()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(args.iters):
                    fn()
                b.record()
                torch.cuda.synchronize()
                times[name][v].append(a.elapsed_time(b) / args.iters * 1e3)
    setvar(-1)
    res = {"tokens": T, "rows": R, "band": args.band, "parity_vs_variant0": parity, "stress": stress, "kernels": {}}
    for name in fns:
        res["kernels"][name] = {}
        for v in variants:
            med, best = statistics.median(times[name][v]), min(times[name][v])
            res["kernels"][name][v] = {"us_median": round(med, 1), "us_best": round(best, 1),
                                       "tflops_median": round(fl[name] / med / 1e6, 1)}
            print(f"{name:7s} variant {v}: {med:8.1f} us median ({best:8.1f} best) = {fl[name] / med / 1e6:7.1f} TFLOP/s",
                  flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    if not all(parity.values()) or any(r["mismatching_launches"] for r in stress.values()):
        sys.exit("PARITY FAILURE between variants")


if __name__ == "__main__":
    main()
