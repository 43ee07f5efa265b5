"""r05 diagnosis: the W8A8 down projection (block-scaled fp8 MFMA) against float64 arithmetic on the same bytes — which
hypothesis about the MFMA's inner arithmetic explains the rare outliers (err up to 2.5e-3 of sum |terms|)?
python tools/fp8a8_outlier.py   (GPU box)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from nanovllm import hip_ops as ops
from oracle import glm4_moe_oracle as O
import fullsize as F
BF = torch.bfloat16
cfg = F.air_cfg()
H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
T = 4096
x = F.randn_bf16(T, H, seed=431)
gw = F.randn_bf16(E, H, seed=432, scale=0.02)
bias = (torch.rand(E, generator=torch.Generator().manual_seed(433)) * 0.2 - 0.1).float().cuda()
idx, w = ops.router_topk(x, gw, bias, k, 1.0)
offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
g = torch.Generator(device="cuda").manual_seed(9)
w13 = torch.empty(E + 1, 2 * I, H, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
w2 = torch.empty(E + 1, H, I, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
q13, s13 = ops.quantize_fp8_rowwise(w13); q2, s2 = ops.quantize_fp8_rowwise(w2)
del w13, w2
R = T * k + T
xq, xs = ops.quantize_fp8_rows(x)
act_q, act_bs = ops.grouped_gemm_gateup_silu_fp8a8(xq, xs, q13, s13, offs, tok, R)
ones = torch.ones(R, dtype=BF, device="cuda")
y = ops.grouped_gemm_down_fp8a8(act_q, act_bs, q2, s2, offs, ones, 0)   # unscaled rows: y = bf16(acc * ws)
torch.cuda.synchronize()
offs_c = offs.cpu()
nblk = I // 32
e = 63
lo, hi = int(offs_c[e]), int(offs_c[e + 1])
rows = torch.arange(lo, min(hi, lo + 256))
aq = act_q[rows.cuda()].cpu().view(torch.float8_e4m3fn)
sb = O.mx_scale_unrecord(act_bs[rows.cuda()].cpu(), nblk)
wq = q2[e].cpu().contiguous().view(torch.float8_e4m3fn)
ws = s2[e].cpu().double()
a64 = aq.to(torch.float32).double().view(-1, nblk, 32)
w64 = wq.to(torch.float32).double().view(H, nblk, 32)
sc = torch.exp2(sb.double() - 127.0)
got = y[rows.cuda()].cpu().double()
def evaluate(a, w_):
    blk = torch.einsum("rbk,nbk->rnb", a, w_)            # per-block sums, float64
    return (blk * sc.unsqueeze(1)).sum(-1) * ws.unsqueeze(0)
exact = evaluate(a64, w64)
ref_bf = exact.float().to(BF).double()
mag = (torch.einsum("rbk,nbk->rnb", a64.abs(), w64.abs()) * sc.unsqueeze(1)).sum(-1) * ws.unsqueeze(0)
err = (got - exact).abs()
ratio = err / mag
top = torch.topk(ratio.flatten(), 8)
print("worst err/mag vs float64 on the same bytes:", [f"{v:.3g}" for v in top.values.tolist()])
sub_a = a64.abs() < 2.0 ** -6      # e4m3 subnormals of the act operand
sub_w = w64.abs() < 2.0 ** -6
hyps = {"flush act subnormals": evaluate(torch.where(sub_a, torch.zeros_like(a64), a64), w64),
        "flush weight subnormals": evaluate(a64, torch.where(sub_w, torch.zeros_like(w64), w64)),
        "flush both": evaluate(torch.where(sub_a, torch.zeros_like(a64), a64), torch.where(sub_w, torch.zeros_like(w64), w64))}
for i in top.indices.tolist()[:5]:
    r, n = divmod(i, H)
    print(f"row {r} col {n}: kernel {got[r, n]:.7g}  float64 {exact[r, n]:.7g} (bf16 {ref_bf[r, n]:.7g})  mag {mag[r, n]:.4g}  "
          + "  ".join(f"{name}: {v[r, n]:.7g}" for name, v in hyps.items()))
    blk = (torch.einsum("bk,bk->b", a64[r], w64[n]) * sc[r]) * ws[n]
    d = got[r, n] - exact[r, n]
    close = torch.topk(-(blk.abs() - abs(d)).abs(), 3)
    print(f"     difference {d:.5g}; block sums closest to it in magnitude: {[(int(b), float(blk[b])) for b in close.indices.tolist()]}; "
          f"scale bytes of the row: min {int(sb[r].min())} max {int(sb[r].max())}; act subnormals in row {int(sub_a[r].sum())}")
    # single terms closest to the difference
    terms = (a64[r] * w64[n] * sc[r].unsqueeze(-1) * ws[n]).flatten()
    ct = torch.topk(-(terms.abs() - abs(d)).abs(), 3)
    print(f"     single terms closest in magnitude: {[(int(j), float(terms[j]), float(a64[r].flatten()[j]), float(w64[n].flatten()[j])) for j in ct.indices.tolist()]}")
print("fraction of elements whose bf16 result equals bf16(float64):", float((got == ref_bf).double().mean()))
# ---- scaled rows (routing weight on bf16(acc * ws)) against the unscaled rows x scale, and run-to-run determinism ----
y_s1 = ops.grouped_gemm_down_fp8a8(act_q, act_bs, q2, s2, offs, scale, E)
y_s2 = ops.grouped_gemm_down_fp8a8(act_q, act_bs, q2, s2, offs, scale, E)
y_u2 = ops.grouped_gemm_down_fp8a8(act_q, act_bs, q2, s2, offs, ones, 0)
torch.cuda.synchronize()
print("unscaled run twice identical:", bool(torch.equal(y, y_u2)), " scaled run twice identical:", bool(torch.equal(y_s1, y_s2)))
nr = int(offs_c[E])   # routed rows
expect = (y[:nr].float() * scale[:nr].float().unsqueeze(-1)).to(BF)
neq = (expect != y_s1[:nr])
print("scaled == bf16(unscaled * scale) on the routed rows:", bool(not neq.any()), " differing elements:", int(neq.sum()), "of", neq.numel())
if neq.any():
    r_, c_ = torch.nonzero(neq)[0].tolist()
    print("  first difference: row", r_, "col", c_, "scaled", float(y_s1[r_, c_]), "expected", float(expect[r_, c_]), "unscaled", float(y[r_, c_]), "scale", float(scale[r_]))
    rows_bad = torch.nonzero(neq.any(dim=1)).flatten()
    print("  rows with differences:", rows_bad.numel(), "first", rows_bad[:10].tolist(), " per-row count (first 5):", neq[rows_bad[:5]].sum(dim=1).tolist())
    g_of = torch.bucketize(rows_bad.cpu(), offs_c[1:E + 1].long(), right=True)
    print("  groups:", sorted(set(g_of.tolist()))[:20], " row offset inside its 256-row block:", [int((r - offs_c[g]) % 256) for r, g in zip(rows_bad[:10].tolist(), g_of[:10].tolist())])
