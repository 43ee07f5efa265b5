"""Debug aid: where does the 64-rows-per-wave prefill attention differ from the default kernel?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
BF = torch.bfloat16
def case(lens_q, lens_k, nh, nkv):
    g = torch.Generator().manual_seed(1)
    hd = 128
    q = torch.randn(sum(lens_q), nh, hd, generator=g).to(BF).cuda()
    k = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF).cuda()
    v = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF).cuda()
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32).cuda()
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32).cuda()
    run = lambda: ops.varlen_attn_prefill(q, k, v, cu_q, cu_k, max(lens_q), hd ** -0.5)
    ops.set_option("fa_rows64", 0); base = run().float().cpu()
    ops.set_option("fa_rows64", 1); out = run().float().cpu()
    ops.set_option("fa_rows64", 0)
    bad = (out - base).abs() > 0.02 + 0.02 * base.abs()
    print(f"case q={lens_q} k={lens_k} nh={nh} nkv={nkv}: bad {int(bad.sum())}/{bad.numel()} nan {int(torch.isnan(out).sum())}")
    if bad.any():
        rows = bad.any(2).any(1).nonzero().flatten().tolist()
        # compress runs
        runs = []; s = rows[0]; p = rows[0]
        for r in rows[1:]:
            if r != p + 1: runs.append((s, p)); s = r
            p = r
        runs.append((s, p))
        print("  bad row runs:", runs[:24])
        print("  bad heads:", bad.any(2).any(0).nonzero().flatten().tolist()[:32])
        cols = bad.any(1).any(0).nonzero().flatten().tolist()
        print("  bad cols:", len(cols), cols[:16])
        r0 = rows[0]; h0 = bad[r0].any(1).nonzero().flatten().tolist()[0]
        print("  sample row", r0, "head", h0, "got", out[r0, h0, :6].tolist(), "want", base[r0, h0, :6].tolist())
for c in [([1], [1], 1, 1), ([17], [17], 1, 1), ([33], [33], 1, 1), ([64], [64], 1, 1), ([200], [200], 1, 1), ([513], [513], 1, 1), ([2048], [2048], 1, 1)]:
    case(*c)
