"""Phase timeline of the fused add-RMSNorm + router + align launch (study tool).

Needs the trace variant of the library:
    python glm4_moe_nano-vllm_amd/build.py --variant _rtrace GLM_ROUTER_TRACE=1
    GLM4MOE_HIP_LIB=glm4_moe_nano-vllm_amd/lib/libglm4moe_hip_rtrace.so python tools/router_trace.py [--tokens 32] [--cold]

Thread 0 of every workgroup stamps the 100 MHz wall clock at ten phase boundaries (moe_misc.hip RT_MARK).  Printed per
phase: when the first / median / last workgroup reached it, relative to the first workgroup's entry.  --cold streams
1 GiB through the caches between launches (the layer's weight stream does that to the gate weights between routers).
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
import torch

from nanovllm import hip_ops as ops

PHASES = ["entry", "inputs loaded", "norm row in LDS", "dots done (wave 0)", "logits drained + met", "token ticket",
          "logits gathered", "selection stored", "global ticket", "bucketing done"]
SLOTS, WGS = 12, 4096


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", type=int, default=32)
    ap.add_argument("--slices", type=int, default=4)
    ap.add_argument("--cold", action="store_true")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lib = ops.load_library()
    tr = lib.glm_debug_router_trace
    tr.restype = ctypes.c_int
    tr.argtypes = [ctypes.c_void_p, ctypes.c_int]
    T, H, E, k = args.tokens, 4096, 64, 8
    g = torch.Generator(device="cuda").manual_seed(3)
    part = torch.randn(args.slices, T, H, device="cuda", generator=g) * 0.5
    res = torch.randn(T, H, device="cuda", generator=g).to(torch.bfloat16)
    nw = (1 + 0.1 * torch.randn(H, device="cuda", generator=g)).to(torch.bfloat16)
    gw = (torch.randn(E, H, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    bias = (torch.rand(E, device="cuda", generator=g) * 0.2 - 0.1).float()
    flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda") if args.cold else None  # 1 GiB
    host = (ctypes.c_ulonglong * (SLOTS * WGS))()
    G = (E + 15) // 16
    nwg = T * G
    per_rep = []
    for rep in range(args.reps + 2):
        if flush is not None:
            flush.add_(1.0)
        part2 = part + 0.0  # fresh producer output, as after the o_proj
        torch.cuda.synchronize()
        assert tr(None, 1) == 0
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.add_rmsnorm_router_topk(ops.PartialSum(part2), res, nw, 1e-5, gw, bias, k, 1.0, align_with_shared=True)
        b.record()
        torch.cuda.synchronize()
        assert tr(ctypes.addressof(host), 0) == 0
        if rep < 2:
            continue
        rows = [[host[w * SLOTS + i] for i in range(10)] for w in range(nwg)]
        t0 = min(r[0] for r in rows)
        ph = []
        for i in range(10):
            v = [(r[i] - t0) / 100.0 for r in rows if r[i]]  # 100 MHz -> us
            ph.append({"n": len(v), "first": min(v), "median": statistics.median(v), "last": max(v)} if v else None)
        per_rep.append({"event_us": a.elapsed_time(b) * 1e3, "phases": ph})
    # median over repetitions of each figure
    print(f"T={T} G={G} workgroups={nwg} slices={args.slices} cold={args.cold}; event time median "
          f"{statistics.median(r['event_us'] for r in per_rep):.1f} us")
    print(f"{'phase':28s} {'n':>5s} {'first':>8s} {'median':>8s} {'last':>8s}   (us after the first workgroup's entry)")
    summary = []
    for i, name in enumerate(PHASES):
        cells = [r["phases"][i] for r in per_rep if r["phases"][i]]
        if not cells:
            continue
        m = {key: statistics.median(c[key] for c in cells) for key in ("n", "first", "median", "last")}
        summary.append({"phase": name, **m})
        print(f"{name:28s} {int(m['n']):5d} {m['first']:8.2f} {m['median']:8.2f} {m['last']:8.2f}")
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        json.dump({"tokens": T, "G": G, "cold": args.cold, "summary": summary,
                   "event_us_median": statistics.median(r["event_us"] for r in per_rep)}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
