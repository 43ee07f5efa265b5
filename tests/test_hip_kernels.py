"""GPU parity tests: every HIP kernel, called through the C-ABI, against the CPU oracle on
seeded inputs and against the golden vectors minted from the reference.

Tolerances (stated per test): elementwise ops with no reduction are BIT-EXACT; ops behind an
fp32 reduction (norms, GEMMs) may differ from the CPU's summation order by one bf16 ulp on a
small fraction of elements; attention is compared with rtol 1.6e-2 / atol 4e-3 (bf16 output,
P rounded to bf16 before P·V as in flash-attn; parity unpinned at that boundary).
"""
import math

import pytest
import torch

from oracle import glm4_moe_oracle as O
from util import assert_close_bf16, load_golden

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from nanovllm import hip_ops
    hip_ops.load_library()
    return hip_ops


def dev(t):
    return t.cuda() if t is not None else None


import contextlib
import os


@contextlib.contextmanager
def largem_variant(v: int):
    """Loop variant of the large-M GEMM for the launches inside the block (library option "largem_variant", consulted at
    every launch; 0 = the plain loop every other variant must equal bit for bit)."""
    from nanovllm import hip_ops
    with hip_ops.option("largem_variant", v):
        yield


@pytest.fixture
def opt_guard():
    """set_option for the rest of a test, restored afterwards: `opt_guard("tinym", 0)`."""
    from nanovllm import hip_ops
    saved = []

    def set_(name, value):
        saved.append((name, hip_ops.set_option(name, value)))
    yield set_
    for name, old in reversed(saved):
        hip_ops.set_option(name, old)


def rnd(*shape, seed=0, scale=1.0, dtype=BF):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype)


def assert_mixed(a, e, rtol, atol, what=""):
    a, e = a.detach().cpu().float(), e.detach().cpu().float()
    assert a.shape == e.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(e.shape)}"
    # a NaN on either side compares False below and would pass unseen (r05: the attention stand-in's softmax over nothing)
    assert bool(torch.isfinite(e).all()), f"{what}: the expected values hold {int((~torch.isfinite(e)).sum())} non-finite elements"
    assert bool(torch.isfinite(a).all()), f"{what}: {int((~torch.isfinite(a)).sum())} non-finite elements"
    err = (a - e).abs()
    lim = atol + rtol * e.abs()
    bad = err > lim
    assert not bool(bad.any()), (f"{what}: {int(bad.sum())}/{a.numel()} outside tolerance, "
                                 f"max err {float(err.max()):.4g}, max ref {float(e.abs().max()):.4g}")


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("T,H", [(1, 128), (5, 4096), (33, 64), (7, 100), (3, 8192), (2, 16384)])
def test_rmsnorm(ops, T, H):
    x, r, w = rnd(T, H, seed=1), rnd(T, H, seed=2), (1 + rnd(H, seed=3, scale=0.1).float()).to(BF)
    eps = 1e-5
    assert_close_bf16(ops.rmsnorm(dev(x), dev(w), eps), O.rms_forward(x, w, eps), 1, 0.02, what="rmsnorm")
    y, ro = ops.add_rmsnorm(dev(x), dev(r), dev(w), eps)
    ye, re = O.add_rms_forward(x, r, w, eps)
    assert torch.equal(ro.cpu(), re), "residual_out must be bit-exact (single fp32 add, one rounding)"
    assert_close_bf16(y, ye, 1, 0.02, what="add_rmsnorm")


def test_elementwise_golden(ops):
    t, m = load_golden("ops")
    eps = m["eps"]
    assert_close_bf16(ops.rmsnorm(dev(t["x"]), dev(t["ln_w"]), eps), t["y_rms"], 1, 0.02, what="rms golden")
    y, r = ops.add_rmsnorm(dev(t["x"]), dev(t["res"]), dev(t["ln_w"]), eps)
    assert torch.equal(r.cpu(), t["r_add"])
    assert_close_bf16(y, t["y_add"], 1, 0.02, what="add_rms golden")
    assert_close_bf16(ops.head_rmsnorm(dev(t["xh"]), dev(t["hn_w"]), eps), t["y_head"], 1, 0.02, what="head golden")
    q, k = ops.rope_neox(dev(t["pos"]), dev(t["q"]), dev(t["k"]), dev(t["cos_sin"]), 64)
    assert torch.equal(q.cpu(), t["q_r"]) and torch.equal(k.cpu(), t["k_r"]), "rope must be bit-exact"
    q, k = ops.rope_neox(dev(t["pos2"]), dev(t["q2"]), dev(t["k2"]), dev(t["cos_sin2"]), 16)
    assert torch.equal(q.cpu(), t["q2_r"]) and torch.equal(k.cpu(), t["k2_r"]), "rope(scalar path) bit-exact"
    assert_close_bf16(ops.silu_and_mul(dev(t["xs"])), t["y_silu"], 1, 0.01, what="silu golden")
    assert_close_bf16(ops.gemm_bf16(dev(t["x"]), dev(t["lin_w"]), dev(t["lin_b"])), t["y_lin"], 1, 0.03, what="linear golden")


def test_head_rmsnorm_strided(ops):
    T, nh, hd = 9, 6, 128
    big = rnd(T, (nh + 4) * hd, seed=5)
    x = big[:, : nh * hd].view(T, nh, hd)  # token stride (nh+4)*hd
    w = (1 + rnd(hd, seed=6, scale=0.1).float()).to(BF)
    assert_close_bf16(ops.head_rmsnorm(dev(big)[:, : nh * hd].view(T, nh, hd), dev(w), 1e-6),
                      O.head_rmsnorm(x, w, 1e-6), 1, 0.02, what="head_rmsnorm strided")
    xs = rnd(4, 3, 24, seed=7)  # hd=24: scalar fallback
    ws = (1 + rnd(24, seed=8, scale=0.1).float()).to(BF)
    assert_close_bf16(ops.head_rmsnorm(dev(xs), dev(ws), 1e-6), O.head_rmsnorm(xs, ws, 1e-6), 1, 0.03, what="head scalar")


def test_silu_and_mul(ops):
    for T, I in [(3, 1408), (5, 100), (1, 8)]:
        x = rnd(T, 2 * I, seed=T, scale=3.0)
        assert_close_bf16(ops.silu_and_mul(dev(x)), O.silu_and_mul(x), 1, 0.01, what=f"silu {T}x{I}")


def test_kv_store(ops):
    N, nkv, hd, nslots = 7, 2, 128, 3 * 256
    qkv = rnd(N, (4 + 2 * nkv) * hd, seed=11)
    k = qkv[:, 4 * hd: (4 + nkv) * hd].view(N, nkv, hd)
    v = qkv[:, (4 + nkv) * hd:].view(N, nkv, hd)
    slots = torch.tensor([5, -1, 700, 256, 0, 767, -1], dtype=torch.int32)
    kc, vc = rnd(3, 256, nkv, hd, seed=12), rnd(3, 256, nkv, hd, seed=13)
    kce, vce = kc.clone(), vc.clone()
    O.store_kvcache(k, v, kce, vce, slots)
    kcd, vcd = dev(kc), dev(vc)
    qd = dev(qkv)
    ops.kv_store(qd[:, 4 * hd: (4 + nkv) * hd].view(N, nkv, hd), qd[:, (4 + nkv) * hd:].view(N, nkv, hd),
                 kcd, vcd, dev(slots))
    assert torch.equal(kcd.cpu(), kce) and torch.equal(vcd.cpu(), vce)


@pytest.mark.parametrize("qknorm", [False, True])
def test_qkv_split_norm_rope_store(ops, qknorm):
    T, nh, nkv, hd, rd = 11, 12, 2, 128, 64
    qkv = rnd(T, (nh + 2 * nkv) * hd, seed=21)
    pos = torch.randint(0, 1024, (T,), generator=torch.Generator().manual_seed(22))
    cs = O.build_cos_sin_cache(rd, 1024, 10000.0)
    qn = (1 + rnd(hd, seed=23, scale=0.1).float()).to(BF) if qknorm else None
    kn = (1 + rnd(hd, seed=24, scale=0.1).float()).to(BF) if qknorm else None
    slots = torch.tensor([3, 258, -1, 0, 511, 100, 101, 102, -1, 300, 7], dtype=torch.int32)
    kc, vc = rnd(2, 256, nkv, hd, seed=25), rnd(2, 256, nkv, hd, seed=26)
    # oracle composition (attention_new.py:329-357 + layers/attention.py:75)
    q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
    q, k, v = q.reshape(T, nh, hd), k.reshape(T, nkv, hd), v.reshape(T, nkv, hd)
    if qknorm:
        q, k = O.head_rmsnorm(q, qn, 1e-5), O.head_rmsnorm(k, kn, 1e-5)
    qe, ke = O.rope_neox(pos, q, k, cs, rd)
    kce, vce = kc.clone(), vc.clone()
    O.store_kvcache(ke, v, kce, vce, slots)
    kcd, vcd = dev(kc), dev(vc)
    qo, ko, vo = ops.qkv_split_norm_rope_store(dev(qkv), nh, nkv, hd, dev(pos), dev(cs), rd, dev(qn), dev(kn),
                                               1e-5, kcd, vcd, dev(slots), True)
    assert torch.equal(vo.cpu(), v), "v pass-through must be bit-exact"
    assert torch.equal(vcd.cpu(), vce), "v cache scatter must be bit-exact"
    if qknorm:  # one fp32 reduction in front of the rotary: allow 1 ulp
        assert_close_bf16(qo, qe, 1, 0.02, what="q")
        assert_close_bf16(ko, ke, 1, 0.02, what="k")
        assert_close_bf16(kcd, kce, 1, 0.02, what="k cache")
    else:
        assert torch.equal(qo.cpu(), qe) and torch.equal(ko.cpu(), ke), "split+rope must be bit-exact"
        assert torch.equal(kcd.cpu(), kce), "k cache scatter must be bit-exact"


# ------------------------------------------------------------------------------------------
def _paged_case(lens, nh, nkv, seed, block_size=256, hd=128, extra_blocks=2):
    g = torch.Generator().manual_seed(seed)
    B = len(lens)
    nb = [max(1, (n + block_size - 1) // block_size) for n in lens]
    total = sum(nb) + extra_blocks
    perm = torch.randperm(total, generator=g)
    maxb = max(nb)
    bt = torch.full((B, maxb), -1, dtype=torch.int32)
    c = 0
    for b in range(B):
        for j in range(nb[b]):
            bt[b, j] = int(perm[c]); c += 1
    kc = torch.randn(total, block_size, nkv, hd, generator=g).to(BF)
    vc = torch.randn(total, block_size, nkv, hd, generator=g).to(BF)
    q = torch.randn(B, nh, hd, generator=g).to(BF)
    return q, kc, vc, torch.tensor(lens, dtype=torch.int32), bt


@pytest.mark.parametrize("lens,nh,nkv", [
    ([300, 256, 5], 4, 2), ([1, 31, 32, 33, 1000], 24, 2), ([0, 64, 0, 700], 8, 8),
    ([2048, 129], 96, 8), ([17], 16, 1)])
@pytest.mark.parametrize("splits", [0, 1, 3])
def test_paged_attn_decode(ops, lens, nh, nkv, splits):
    q, kc, vc, cl, bt = _paged_case(lens, nh, nkv, seed=len(lens) * 7 + nh)
    scale = 128 ** -0.5
    ref = O.attn_decode_paged(q, kc, vc, cl, bt, scale)
    out = ops.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(cl), dev(bt), scale, num_splits=splits)
    assert torch.isfinite(out.float()).all()
    for b, n in enumerate(lens):
        if n == 0:
            assert float(out[b].float().abs().max()) == 0.0, "context_len 0 must give a zero row"
    assert_mixed(out, ref, 1.6e-2, 4e-3, f"paged decode lens={lens} splits={splits}")


@pytest.mark.parametrize("lens,nh,nkv,splits", [([300, 256, 5, 33], 4, 2, 0), ([1, 31, 32, 33, 1000], 24, 2, 3),
                                                 ([2048, 129], 96, 8, 0), ([17], 16, 1, 2), ([513, 512], 12, 1, 5),
                                                 # one split over many key groups: the ring starts before the new row is
                                                 # stored (513 = the new key alone in the last group; 544 = 17 groups, the
                                                 # shortest range that takes that path; 512 = 16 groups, the longest that
                                                 # does not)
                                                 ([1500, 513, 544, 512], 12, 1, 1), ([4096, 3000], 16, 2, 1)])
def test_paged_attn_decode_from_qkv(ops, lens, nh, nkv, splits):
    """One launch (rotary + K/V cache store + attention from the packed QKV rows) == split/rope/store followed by
    the paged attention, bit for bit: outputs and cache contents."""
    _, kc, vc, cl, bt = _paged_case(lens, nh, nkv, seed=len(lens) * 11 + nh)
    B, hd, bs, rd = len(lens), 128, 256, 64
    g = torch.Generator().manual_seed(1234 + nh)
    qkv = (torch.randn(B, (nh + 2 * nkv) * hd, generator=g) * 1.5).to(BF)
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int64)
    slots = torch.tensor([int(bt[b, (n - 1) // bs]) * bs + (n - 1) % bs for b, n in enumerate(lens)], dtype=torch.int32)
    inv = 1.0 / (1e6 ** (torch.arange(0, rd, 2, dtype=torch.float) / rd))
    fr = torch.outer(torch.arange(4096, dtype=torch.float), inv)
    cs = torch.cat((fr.cos(), fr.sin()), -1).unsqueeze(1).contiguous()
    scale = hd ** -0.5
    kc0, vc0, kc1, vc1 = dev(kc), dev(vc), dev(kc), dev(vc)
    q, _, _ = ops.qkv_split_norm_rope_store(dev(qkv), nh, nkv, hd, dev(pos), dev(cs), rd, None, None, 1e-5,
                                            kc0, vc0, dev(slots), False)
    out0 = ops.paged_attn_decode(q, kc0, vc0, dev(cl), dev(bt), scale, num_splits=splits)
    out1 = ops.paged_attn_decode_qkv(dev(qkv), nh, nkv, hd, dev(pos), dev(cs), rd, dev(slots), kc1, vc1, dev(cl),
                                     dev(bt), scale, num_splits=splits)
    assert torch.equal(kc0, kc1) and torch.equal(vc0, vc1), "cache contents differ"
    assert not torch.equal(kc1.cpu(), kc), "the new rows must have been written"
    assert torch.equal(out0, out1)


@pytest.mark.parametrize("lens,nh,nkv,splits,ns,with_bias", [([300, 256, 5, 33], 4, 2, 0, 2, True),
                                                              ([2048, 129], 96, 8, 0, 2, True), ([17], 16, 1, 2, 3, False),
                                                              ([513, 512, 1], 12, 1, 5, 4, True),
                                                              ([1500, 544, 512], 12, 1, 1, 2, True)])
def test_paged_attn_decode_from_qkv_partial_sums(ops, lens, nh, nkv, splits, ns, with_bias):
    """The fused decode attention fed by fp32 K-slice partial sums of the QKV projection + bias == the same launch fed by
    the bf16 rows bf16(slice 0 + slice 1 + ... + bias), bit for bit (outputs and cache contents): the reduction on load
    is the linear layer's own output rounding (linear.py:74-75)."""
    _, kc, vc, cl, bt = _paged_case(lens, nh, nkv, seed=len(lens) * 13 + nh)
    B, hd, bs, rd = len(lens), 128, 256, 64
    W = (nh + 2 * nkv) * hd
    g = torch.Generator().manual_seed(4321 + nh)
    part = torch.randn(ns, B, W, generator=g) * (1.5 / ns ** 0.5)
    bias = (torch.randn(W, generator=g) * 0.1).to(BF) if with_bias else None
    tot = part[0].clone()
    for i in range(1, ns):
        tot += part[i]
    if with_bias:
        tot += bias.float()
    qkv = tot.to(BF)
    pos = torch.tensor([n - 1 for n in lens], dtype=torch.int64)
    slots = torch.tensor([int(bt[b, (n - 1) // bs]) * bs + (n - 1) % bs for b, n in enumerate(lens)], dtype=torch.int32)
    inv = 1.0 / (1e6 ** (torch.arange(0, rd, 2, dtype=torch.float) / rd))
    fr = torch.outer(torch.arange(4096, dtype=torch.float), inv)
    cs = torch.cat((fr.cos(), fr.sin()), -1).unsqueeze(1).contiguous()
    scale = hd ** -0.5
    kc0, vc0, kc1, vc1 = dev(kc), dev(vc), dev(kc), dev(vc)
    out0 = ops.paged_attn_decode_qkv(dev(qkv), nh, nkv, hd, dev(pos), dev(cs), rd, dev(slots), kc0, vc0, dev(cl),
                                     dev(bt), scale, num_splits=splits)
    out1 = ops.paged_attn_decode_qkv(ops.PartialSum(dev(part)), nh, nkv, hd, dev(pos), dev(cs), rd, dev(slots), kc1, vc1,
                                     dev(cl), dev(bt), scale, num_splits=splits,
                                     qkv_bias=dev(bias) if with_bias else None)
    assert torch.equal(kc0, kc1) and torch.equal(vc0, vc1), "cache contents differ"
    assert torch.equal(out0, out1)


def test_qkv_projection_as_partial_sums_matches_one_pass(ops):
    """gemm_bf16_partial at the QKV shape (2 slices) + bias, reduced as the attention kernel does, against the one-pass
    gemm_bf16: the same bf16 values up to fp32 summation order (<= 1 ulp on a few per cent)."""
    M, K, N = 32, 4096, 14336
    x, w = dev(rnd(M, K, seed=401)), dev(rnd(N, K, seed=402, scale=0.02))
    b = dev(rnd(N, seed=403, scale=0.02))
    one = ops.gemm_bf16(x, w, b)
    ps = ops.gemm_bf16_partial(x, w, 2)
    assert ps.part.shape[0] == 2
    red = (ps.part[0] + ps.part[1] + b.float()).to(BF)
    assert_close_bf16(red, one, 1, 0.05, atol=2 ** -9, what="QKV partial sums vs one pass")


def test_paged_attn_decode_spiky_scores(ops):
    """Force the online-softmax rescale: one key far above the rest late in the sequence."""
    q, kc, vc, cl, bt = _paged_case([777], 12, 1, seed=99)
    blk, off = int(bt[0, 2]), 600 - 512
    kc[blk, off] = (q[0, 3].float() * 4.0).to(BF)  # huge score for head 3 at key 600
    scale = 128 ** -0.5
    ref = O.attn_decode_paged(q, kc, vc, cl, bt, scale)
    for s in (1, 2):
        out = ops.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(cl), dev(bt), scale, num_splits=s)
        assert_mixed(out, ref, 1.6e-2, 4e-3, "spiky")


# ------------------------------------------------------------------------------------------
def _check_router(ops, x, gw, bias, k, rsf, what, allow_flip_rows=0.0):
    idx, w, logits = ops.router_topk(dev(x), dev(gw), dev(bias), k, rsf, return_logits=True)
    idx, w, logits = idx.cpu(), w.cpu(), logits.cpu()
    ridx, rw = O.router(x, gw, bias, k, rsf)
    rlogits, rscores, rchoice = O.router_scores(x, gw, bias)
    T = x.shape[0]
    assert bool((idx[:, 1:] > idx[:, :-1]).all()), "indices must be strictly ascending per row"
    n_unexplained, n_flip = 0, 0
    for t in range(T):
        same = set(idx[t].tolist()) == set(ridx[t].tolist())
        if same:
            order = torch.argsort(ridx[t])
            if not torch.equal(w[t], rw[t][order]):
                # weights may differ by one ulp only if a selected score's sigmoid rounded differently
                assert_close_bf16(w[t], rw[t][order], 1, 1.0, what=f"{what} row {t} weights")
            continue
        logits_differ = not torch.equal(logits[t], rlogits[t])
        sel = torch.zeros(gw.shape[0], dtype=torch.bool); sel[ridx[t]] = True
        tie = float(rchoice[t][sel].min()) == float(rchoice[t][~sel].max())
        if logits_differ or tie:
            n_flip += 1
        else:
            n_unexplained += 1
    assert n_unexplained == 0, f"{what}: {n_unexplained} rows with a different expert set and identical logits/no tie"
    assert n_flip <= allow_flip_rows * T, f"{what}: {n_flip}/{T} rows differ through rounding flips/ties"
    return logits, rlogits


def test_router_golden(ops):
    t, meta = load_golden("router")
    for tag, m in meta.items():
        x, gw, bias = t[f"{tag}.x"], t[f"{tag}.gate_w"], t[f"{tag}.bias"]
        if tag == "ties":
            # bf16-quantised scores with zero bias tie; torch.topk's choice among equals is
            # unspecified, ours is the lowest id: the selected VALUES must agree exactly.
            idx, w = ops.router_topk(dev(x), dev(gw), dev(bias), m["k"], m["rsf"])
            _, rscores, rchoice = O.router_scores(x, gw, bias)
            got = torch.sort(rchoice.gather(1, idx.cpu()), dim=1)[0]
            exp = torch.sort(rchoice.gather(1, t[f"{tag}.idx"]), dim=1)[0]
            assert torch.equal(got, exp), "tie case: selected score multiset differs"
            assert torch.equal(torch.sort(w.cpu().float(), dim=1)[0], torch.sort(t[f"{tag}.w"].float(), dim=1)[0])
        else:
            # at most one row per case may differ, and only where the bf16 logit itself sits on a
            # rounding boundary (fp32 summation order; _check_router proves that is the cause)
            _check_router(ops, x, gw, bias, m["k"], m["rsf"], f"router golden {tag}",
                          allow_flip_rows=1.0 / x.shape[0])


def test_router_boundary_ties_golden(ops):
    """Exact ties ON the top-k boundary, every row (tests/golden/router_ties, minted from the reference): gate rows come
    in identical triples, so the 8th and 9th score are equal and two of three equal candidates must be taken.  torch.topk
    leaves that pick unspecified; the product takes the lowest expert ids.  Checked per row: (1) the selected score
    multiset and the weight multiset equal the reference's (rows whose bf16 logits equal the oracle's: all but rounding
    flips of the fp32 summation order, and triples flip together), (2) the set is exactly {everything above the boundary
    value} + {the lowest ids among the equals}, from the product's own logits — on the rank-computation path (E = 64),
    the arg-max-round path (E = 128), and on glm_router_select fed with the same logits."""
    t, meta = load_golden("router_ties")
    for tag, m in meta.items():
        x, gw, bias, k, E = t[f"{tag}.x"], t[f"{tag}.gate_w"], t[f"{tag}.bias"], m["k"], m["E"]
        idx, w, logits = ops.router_topk(dev(x), dev(gw), dev(bias), k, m["rsf"], return_logits=True)
        idx2, w2 = ops.router_select(logits, dev(bias), k, m["rsf"])
        assert torch.equal(idx2, idx) and torch.equal(w2, w), f"{tag}: glm_router_select differs from the fused kernel"
        idx, w, logits = idx.cpu(), w.cpu(), logits.cpu()
        rlogits, _, rchoice = O.router_scores(x, gw, bias)
        choice = logits.sigmoid() + bias.unsqueeze(0)      # R2, R3 on the product's own logits
        n_boundary = n_same_logits = 0
        for r in range(x.shape[0]):
            kth = torch.sort(choice[r], descending=True)[0][k - 1]
            above = [e for e in range(E) if choice[r, e] > kth]
            tied = [e for e in range(E) if choice[r, e] == kth]
            n_boundary += len(tied) > k - len(above)
            want = sorted(above + tied[: k - len(above)])
            assert idx[r].tolist() == want, f"{tag} row {r}: {idx[r].tolist()} != lowest-id pick {want}"
            if torch.equal(logits[r], rlogits[r]):
                n_same_logits += 1
                got = torch.sort(rchoice[r][idx[r]])[0]
                exp = torch.sort(rchoice[r][t[f"{tag}.idx"][r]])[0]
                assert torch.equal(got, exp), f"{tag} row {r}: selected score multiset differs from the reference's"
                assert torch.equal(torch.sort(w[r].float())[0], torch.sort(t[f"{tag}.w"][r].float())[0]), \
                    f"{tag} row {r}: weight multiset differs from the reference's"
        assert n_boundary == x.shape[0], f"{tag}: only {n_boundary} rows tie on the boundary"
        assert n_same_logits >= x.shape[0] - 2, f"{tag}: {x.shape[0] - n_same_logits} rows with logit rounding flips"


def test_router_air_shape(ops):
    """E=64, k=8, H=4096 (config 3 dims), 256 tokens: bit-exact expert sets except rows where the
    bf16 logit itself rounded differently (fp32 summation order) — those must stay below 3 %."""
    T, H, E, k = 256, 4096, 64, 8
    x = rnd(T, H, seed=31)
    gw = rnd(E, H, seed=32, scale=0.02)
    bias = (torch.rand(E, generator=torch.Generator().manual_seed(33)) * 0.2 - 0.1).float()
    logits, rlogits = _check_router(ops, x, gw, bias, k, 1.0, "router air", allow_flip_rows=0.03)
    assert float((logits != rlogits).float().mean()) < 2e-3, "too many logit rounding flips"


@pytest.mark.parametrize("T,H,E,k", [(2048, 4096, 64, 8), (1024, 1024, 128, 8), (1500, 512, 16, 4), (1030, 768, 24, 2)])
def test_router_prefill_size_gemm_path(ops, T, H, E, k):
    """From ROUTER_GEMM_MIN_ROWS rows on the logits are a GEMM and glm_router_select does R2..R7: expert sets bit-exact
    vs the oracle except rows whose bf16 logit itself rounded differently (proven by _check_router), and the fused
    norm + router + bucketing entry point returns exactly norm -> router -> align."""
    assert ops.router_by_gemm(T, E)
    x = rnd(T, H, seed=501)
    gw = rnd(E, H, seed=502, scale=0.02)
    bias = (torch.rand(E, generator=torch.Generator().manual_seed(503)) * 0.2 - 0.1).float()
    logits, rlogits = _check_router(ops, x, gw, bias, k, 1.0, f"router GEMM path T={T}", allow_flip_rows=0.03)
    assert float((logits != rlogits).float().mean()) < 3e-3, "too many logit rounding flips"
    res, nw = rnd(T, H, seed=504), (1 + 0.1 * rnd(H, seed=505).float()).to(BF)
    y, ro, idx, w, al = ops.add_rmsnorm_router_topk(dev(x), dev(res), dev(nw), 1e-5, dev(gw), dev(bias), k, 1.0,
                                                     align_with_shared=True)
    y2, ro2 = ops.add_rmsnorm(dev(x), dev(res), dev(nw), 1e-5)
    idx2, w2 = ops.router_topk(y2, dev(gw), dev(bias), k, 1.0)
    assert torch.equal(y, y2) and torch.equal(ro, ro2) and torch.equal(idx, idx2) and torch.equal(w, w2)
    _check_align(*al, idx.cpu(), w.cpu(), E, True)
    # the selection itself is the decode kernel's: same logits in -> same routing out
    lg = ops.gemm_bf16(y2, dev(gw), None)
    idx3, w3 = ops.router_select(lg, dev(bias), k, 1.0)
    assert torch.equal(idx3, idx2) and torch.equal(w3, w2)


def test_router_select_equals_decode_kernel_on_its_logits(ops):
    """glm_router_select on the logits the token-per-workgroup kernel reports == that kernel's own routing (ties incl.)."""
    for T, H, E, k, seed in ((37, 1024, 64, 8, 1), (64, 512, 128, 8, 2), (20, 256, 8, 2, 3)):
        x = rnd(T, H, seed=600 + seed)
        gw = rnd(E, H, seed=610 + seed, scale=0.05)
        bias = torch.zeros(E) if seed == 3 else (torch.rand(E, generator=torch.Generator().manual_seed(seed)) * 0.2 - 0.1).float()
        idx, w, logits = ops.router_topk(dev(x), dev(gw), dev(bias), k, 2.5, return_logits=True)
        idx2, w2 = ops.router_select(logits, dev(bias), k, 2.5)
        assert torch.equal(idx, idx2) and torch.equal(w, w2)


def _check_align(offs, tok, scale, pos, idx, w, E, shared):
    """moe_align's contract: groups in ascending expert id, every (token, expert) pair exactly once, pair_pos ↔ rows."""
    T, k = idx.shape
    offs, tok, scale, pos = [a.cpu() for a in (offs, tok, scale, pos)]
    counts = torch.bincount(idx.flatten(), minlength=E)
    exp_offs = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    assert torch.equal(offs[: E + 1].long(), exp_offs)
    for e in range(E):
        seg = range(int(offs[e]), int(offs[e + 1]))
        toks = sorted(int(tok[r]) for r in seg)
        assert toks == sorted(torch.where((idx == e).any(dim=1))[0].tolist())
    flat_pos = pos.view(T, k)
    for t in range(T):
        for j in range(k):
            r = int(flat_pos[t, j])
            assert int(offs[int(idx[t, j])]) <= r < int(offs[int(idx[t, j]) + 1])
            assert int(tok[r]) == t and scale[r] == w[t, j]
    assert len(set(pos.tolist())) == T * k
    if shared:
        assert int(offs[E + 1]) == T * k + T
        assert tok[T * k:].tolist() == list(range(T))
        assert bool((scale[T * k:].float() == 1.0).all())


@pytest.mark.parametrize("T,k,E,shared", [(4096, 8, 64, True), (2100, 8, 128, False), (16384, 1, 4, True), (9000, 2, 16, True)])
def test_moe_align_parallel(ops, T, k, E, shared):
    """Prefill-size bucketing (count / offsets / scatter over chunks of 8192 pairs): moe_align's contract, incl. an expert
    nobody picked, a ragged last chunk and the shared group."""
    assert T * k >= ops.ALIGN_PARALLEL_MIN_PAIRS
    g = torch.Generator().manual_seed(T + k + E)
    pool = max(k, E - 1)  # expert E-1 is never selected
    idx = torch.stack([torch.randperm(pool, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    offs, tok, scale, pos = ops.moe_align(dev(idx), dev(w), E, shared)
    offs, tok, scale, pos = [a.cpu() for a in (offs, tok, scale, pos)]
    counts = torch.bincount(idx.flatten(), minlength=E)
    assert torch.equal(offs[: E + 1].long(), torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]))
    P = T * k
    assert sorted(pos.tolist()) == list(range(P)), "pair_pos must be a permutation of the routed rows"
    flat_idx, flat_w = idx.flatten(), w.flatten()
    rows_expert = torch.bucketize(pos.long(), offs[1: E + 1].long(), right=True)  # group of each pair's row
    assert torch.equal(rows_expert, flat_idx), "a pair's row must lie in its expert's group"
    assert torch.equal(tok[pos.long()].long(), torch.arange(P) // k) and torch.equal(scale[pos.long()], flat_w)
    if shared:
        assert int(offs[E + 1]) == P + T and tok[P:].tolist() == list(range(T)) and bool((scale[P:].float() == 1.0).all())


def test_moe_align(ops):
    T, k, E = 37, 4, 16
    g = torch.Generator().manual_seed(41)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = rnd(T, k, seed=42).abs()
    for shared in (True, False):
        _check_align(*ops.moe_align(dev(idx), dev(w), E, shared), idx, w, E, shared)


def _moe_hip(ops, x, gate_w, bias, k, rsf, w13, w2, sw13, sw2, topk=None):
    xd = dev(x)
    if topk is None:
        idx, w = ops.router_topk(xd, dev(gate_w), dev(bias), k, rsf)
    else:
        idx, w = dev(topk[0]), dev(topk[1])
    E = w13.shape[0]
    T = x.shape[0]
    W13 = dev(torch.cat([w13, sw13.unsqueeze(0)]).contiguous())
    W2 = dev(torch.cat([w2, sw2.unsqueeze(0)]).contiguous())
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu(xd, W13, offs, tok, R)
    y = ops.grouped_gemm_down(act, W2, offs, scale, E)
    return ops.moe_combine(y, pos, T, k, T * k), (idx, w, offs, tok, scale, pos, act, y)


def test_moe_golden(ops):
    t, cfg = load_golden("moe")
    E = cfg["n_routed_experts"]
    w13 = torch.stack([t[f"experts.{e}.gate_up_proj.weight"] for e in range(E)])
    w2 = torch.stack([t[f"experts.{e}.down_proj.weight"] for e in range(E)])
    out, (idx, w, *_rest) = _moe_hip(ops, t["x"], t["gate.weight"], t["gate.e_score_correction_bias"],
                                     cfg["num_experts_per_tok"], cfg["routed_scaling_factor"], w13, w2,
                                     t["shared_experts.gate_up_proj.weight"], t["shared_experts.down_proj.weight"])
    gi = torch.sort(t["idx"], dim=1)[0]
    assert torch.equal(idx.cpu(), gi), "golden expert sets must match bit-exactly"
    assert_mixed(out, t["y"], 2 ** -6, 2 ** -6 * float(t["y"].float().abs().max()), "moe golden")


@pytest.mark.parametrize("T,E,k,H,I", [(32, 64, 8, 512, 128), (5, 8, 2, 256, 192), (130, 4, 2, 128, 64), (3, 16, 4, 1024, 352)])
def test_moe_block_vs_oracle(ops, T, E, k, H, I):
    """Expert arithmetic isolated from the router: same (idx, w) on both sides."""
    x = rnd(T, H, seed=51)
    w13, w2 = rnd(E, 2 * I, H, seed=52, scale=H ** -0.5), rnd(E, H, I, seed=53, scale=I ** -0.5)
    sw13, sw2 = rnd(2 * I, H, seed=54, scale=H ** -0.5), rnd(H, I, seed=55, scale=I ** -0.5)
    g = torch.Generator().manual_seed(56)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    ref = O.moe(x, None, None, k, 1.0, w13, w2, sw13, sw2, topk=(idx, w))
    out, (_, _, offs, tok, scale, pos, act, y) = _moe_hip(ops, x, None, None, k, 1.0, w13, w2, sw13, sw2, topk=(idx, w))
    # per-row check of the two GEMMs against the per-expert oracle MLP
    offs, tok = offs.cpu(), tok.cpu()
    W13 = torch.cat([w13, sw13.unsqueeze(0)])
    for e in range(E + 1):
        rows = slice(int(offs[e]), int(offs[e + 1]))
        if rows.start == rows.stop:
            continue
        xe = x[tok[rows].long()]
        a_ref = O.silu_and_mul(O.linear(xe, W13[e]))
        assert_close_bf16(act[rows], a_ref, 2, 0.06, atol=1e-3, what=f"act expert {e}")
    assert_mixed(out, ref, 2 ** -6, 2 ** -6 * float(ref.float().abs().max()), "moe block")


def test_moe_combine_order_is_reference_order(ops):
    """bf16 accumulation in ascending expert id, then + shared: bit-exact given the same rows."""
    T, k, H = 9, 4, 256
    R = T * k + T
    y = rnd(R, H, seed=61, scale=3.0)
    g = torch.Generator().manual_seed(62)
    pos = torch.stack([torch.randperm(T * k, generator=g)[:k] for _ in range(T)])
    # make rows unique across tokens
    pos = torch.randperm(T * k, generator=g).view(T, k).to(torch.int32)
    out = ops.moe_combine(dev(y), dev(pos.flatten().contiguous()), T, k, T * k).cpu()
    exp = torch.zeros(T, H, dtype=BF)
    for t in range(T):
        acc = torch.zeros(H, dtype=BF)
        for p in sorted(pos[t].tolist()):
            acc = acc + y[p]
        exp[t] = acc + y[T * k + t]
    assert torch.equal(out, exp)


@pytest.mark.parametrize("T,k,H,drop", [(9, 4, 256, False), (32, 8, 4096, False), (5, 2, 8192, True), (3, 8, 16384, False)])
def test_moe_combine_fused_with_next_add_rmsnorm(ops, T, k, H, drop):
    """One launch == moe_combine followed by add_rmsnorm(combined, residual), bit for bit."""
    from nanovllm.hip_ops import DeferredCombine
    R = T * k + T
    y = rnd(R, H, seed=161, scale=2.0)
    g = torch.Generator().manual_seed(162)
    pos = torch.randperm(T * k, generator=g).view(T, k).to(torch.int32)
    if drop:
        pos[0, 1] = -1  # a dropped pair contributes nothing
    res = rnd(T, H, seed=163)
    nw = rnd(H, seed=164) * 0.1 + 1.0
    yd, pd = dev(y), dev(pos.flatten().contiguous())
    comb = ops.moe_combine(yd, pd, T, k, T * k)
    y0, r0 = ops.add_rmsnorm(comb, dev(res), dev(nw), 1e-5)
    dc = DeferredCombine(yd, pd, T, k, T * k, (T, H))
    y1, r1 = ops.moe_combine_add_rmsnorm(dc, dev(res), dev(nw), 1e-5)
    assert torch.equal(dc.materialize(), comb)
    assert torch.equal(r0, r1) and torch.equal(y0, y1)


# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K,bias", [
    (1, 64, 128, False), (5, 72, 4096, True), (32, 4096, 1408, False), (33, 512, 176, True),
    (32, 14336, 512, True), (32, 1024, 12288, False), (64, 256, 704, False), (7, 4096, 4096, True),
    (16, 151552 // 8, 256, False), (32, 14336, 4096, False), (9, 12288, 1152, True)])
def test_gemm_bf16_small_m(ops, M, N, K, bias):
    x = rnd(M, K, seed=71)
    w = rnd(N, K, seed=72, scale=K ** -0.5)
    b = rnd(N, seed=73) if bias else None
    ref = O.linear(x, w, b)
    out = ops.gemm_bf16(dev(x), dev(w), dev(b))
    # outputs are O(1); results that cancel to ~0 are compared absolutely (atol = 2^-9)
    assert_close_bf16(out, ref, 1, 0.05, atol=2 ** -9, what=f"gemm {M}x{N}x{K}")


# ---- single-sequence decode: the GEMV body (csrc/gemm_tinym.hip), T <= 2 -----------------------------------------------
@pytest.mark.parametrize("M,N,K,bias", [
    (1, 14336, 4096, True), (2, 14336, 4096, True), (1, 4096, 12288, False), (2, 4096, 12288, False),
    (1, 4096, 1408, False), (2, 64, 136, True), (1, 151552 // 4, 4096, False), (2, 2052, 2056, True), (1, 8, 8, False)])
def test_gemm_tinym_dense(ops, opt_guard, M, N, K, bias):
    """glm_gemm_bf16 at M <= 2 runs the GEMV body: vs the oracle (<= 1 bf16 ulp: fp32 summation order) and vs the small-M
    MFMA kernel the same call takes with GLM4MOE_TINYM=0 (two fp32 evaluations of the same sum: <= 1 ulp apart)."""
    x = rnd(M, K, seed=271)
    w = rnd(N, K, seed=272, scale=K ** -0.5)
    b = rnd(N, seed=273) if bias else None
    ref = O.linear(x, w, b)
    out = ops.gemm_bf16(dev(x), dev(w), dev(b))
    assert_close_bf16(out, ref, 1, 0.05, atol=2 ** -9, what=f"tinym gemm {M}x{N}x{K}")
    opt_guard("tinym", 0)
    old = ops.gemm_bf16(dev(x), dev(w), dev(b))
    assert_close_bf16(out, old, 1, 0.05, atol=2 ** -9, what=f"tinym vs small-M {M}x{N}x{K}")


@pytest.mark.parametrize("T,E,k,H,I", [(1, 64, 8, 4096, 1408), (2, 64, 8, 4096, 1408), (2, 8, 2, 256, 96), (1, 128, 8, 512, 176)])
def test_grouped_gemms_tinym(ops, opt_guard, T, E, k, H, I):
    """Grouped gate|up + SiLU.mul and down at T <= 2 (the few-token entry points): every row against the oracle's expert
    arithmetic, and against the small-M kernels of the same call with GLM4MOE_TINYM=0."""
    x = rnd(T, H, seed=281)
    gw = rnd(E, H, seed=282, scale=H ** -0.5)
    bias = (torch.rand(E, generator=torch.Generator().manual_seed(283)) * 0.2 - 0.1).float()
    w13 = rnd(E + 1, 2 * I, H, seed=284, scale=H ** -0.5)
    w2 = rnd(E + 1, H, I, seed=285, scale=I ** -0.5)
    idx, w = ops.router_topk(dev(x), dev(gw), dev(bias), k, 1.0)
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    R = T * k + T
    w13d, w2d = dev(w13), dev(w2)
    act = ops.grouped_gemm_gateup_silu(dev(x), w13d, offs, tok, R, max_rows_per_group=T)
    y = ops.grouped_gemm_down(act, w2d, offs, scale, E, max_rows_per_group=T)
    opt_guard("tinym", 0)
    act0 = ops.grouped_gemm_gateup_silu(dev(x), w13d, offs, tok, R, max_rows_per_group=T)
    y0 = ops.grouped_gemm_down(act, w2d, offs, scale, E, max_rows_per_group=T)   # same input rows as the GEMV body's
    torch.cuda.synchronize()
    assert_close_bf16(act, act0, 3, 0.06, atol=2 ** -7, what="tinym act vs small-M")
    assert_close_bf16(y, y0, 1, 0.05, atol=2 ** -9, what="tinym y vs small-M")
    offs_c, tok_c, sc_c = offs.cpu(), tok.cpu(), scale.cpu()
    for g in range(E + 1):
        rows = slice(int(offs_c[g]), int(offs_c[g + 1]))
        if rows.stop == rows.start:
            continue
        a_ref = O.silu_and_mul(O.linear(x[tok_c[rows].long()], w13[g]))
        assert_close_bf16(act[rows].cpu(), a_ref, 3, 0.06, atol=2 ** -7, what=f"tinym act group {g}")
        y_ref = O.linear(act[rows].cpu(), w2[g])
        if g < E:
            y_ref = (y_ref.float() * sc_c[rows].float().unsqueeze(1)).to(BF)
        assert_close_bf16(y[rows].cpu(), y_ref, 1, 0.05, atol=2 ** -9, what=f"tinym y group {g}")


@pytest.mark.parametrize("T,k,H,N,bias,drop", [(1, 8, 4096, 14336, True, False), (1, 4, 1024, 264, False, True),
                                               (1, 2, 256, 64, True, False), (1, 8, 2048, 4096, True, True)])
def test_combine_norm_qkv_one_launch(ops, T, k, H, N, bias, drop):
    """glm_gemm_bf16_combine_norm == glm_moe_combine_add_rmsnorm followed by glm_gemm_bf16, bit for bit (one token)."""
    from nanovllm.hip_ops import DeferredCombine
    R = T * k + T
    y = rnd(R, H, seed=291, scale=2.0)
    g = torch.Generator().manual_seed(292)
    pos = torch.randperm(T * k, generator=g).view(T, k).to(torch.int32)
    if drop:
        pos[0, 1] = -1
    res = rnd(T, H, seed=293)
    nw = rnd(H, seed=294) * 0.1 + 1.0
    w = rnd(N, H, seed=295, scale=H ** -0.5)
    b = rnd(N, seed=296) if bias else None
    yd, pd = dev(y), dev(pos.flatten().contiguous())
    dc = DeferredCombine(yd, pd, T, k, T * k, (T, H))
    xn, r0 = ops.moe_combine_add_rmsnorm(dc, dev(res), dev(nw), 1e-5)
    q0 = ops.gemm_bf16(xn, dev(w), dev(b))
    assert ops.can_combine_norm_gemm(dc, H)
    for rep in range(3):  # a late weight prefetch or a race on the LDS row would show as a rare mismatch
        q1, r1 = ops.combine_add_rmsnorm_gemm(dc, dev(res), dev(nw), 1e-5, dev(w), dev(b))
        assert torch.equal(r0, r1), "residual_out differs"
        assert torch.equal(q0, q1), "projection differs from norm launch + GEMV launch"
    assert_close_bf16(q1, O.linear(xn.cpu(), w, b), 1, 0.05, atol=2 ** -9, what="fused qkv vs oracle")


@pytest.mark.parametrize("K,H,E,k,shared", [(12288, 4096, 64, 8, True), (6144, 2048, 128, 8, False), (4096, 2048, 16, 4, True),
                                            (12288, 4096, 72, 3, True)])
def test_oproj_router_tail_one_launch(ops, K, H, E, k, shared):
    """glm_gemm_bf16_router_tail (one token: o_proj GEMV whose last workgroup carries on with add-RMSNorm + router +
    bucketing) == glm_gemm_bf16 followed by glm_add_rmsnorm_router_topk_align, bit for bit; new inputs every repetition
    and unrelated launches in between (a stale line of the handed-over row, or a ticket left non-zero, would show)."""
    wo = dev(rnd(H, K, seed=311, scale=K ** -0.5))
    nw = dev(rnd(H, seed=312) * 0.1 + 1.0)
    gw = dev(rnd(E, H, seed=313, scale=H ** -0.5))
    bias = dev((torch.rand(E, generator=torch.Generator().manual_seed(314)) * 0.2 - 0.1).float())
    for rep in range(6):
        x, res = dev(rnd(1, K, seed=320 + rep)), dev(rnd(1, H, seed=330 + rep))
        o0 = ops.gemm_bf16(x, wo, None)
        y0, r0, i0, w0, al0 = ops.add_rmsnorm_router_topk(o0, res, nw, 1e-5, gw, bias, k, 2.5, align_with_shared=shared)
        y1, r1, i1, w1, al1 = ops.gemm_add_rmsnorm_router_topk_align(x, wo, res, nw, 1e-5, gw, bias, k, 2.5, shared)
        ops.rmsnorm(dev(rnd(64, 4096, seed=rep)), dev(rnd(4096, seed=7)), 1e-5)
        assert torch.equal(y0, y1) and torch.equal(r0, r1), f"rep {rep}: norm outputs differ"
        assert torch.equal(i0, i1) and torch.equal(w0, w1), f"rep {rep}: routing differs"
        for a, b, name in zip(al0, al1, ("offsets", "sorted_token", "sorted_scale", "pair_pos")):
            assert torch.equal(a, b), f"rep {rep}: {name} differs"
        _check_align(*al1, i1.cpu(), w1.cpu(), E, shared)
    head = next(iter(ops._tail_ws.values()))[:8].view(torch.int32)
    assert int(head.abs().sum()) == 0, "both counters must be back at zero"


@pytest.mark.parametrize("K,H,E,k,I", [(12288, 4096, 128, 8, 1408), (12288, 4096, 64, 8, 1408), (4096, 2048, 16, 4, 1024),
                                        (6144, 2048, 72, 3, 1536)])
def test_decode_mlp_engine_one_launch(ops, K, H, E, k, I):
    """r05, BASELINE configs[1]: glm_decode_mlp_engine — o_proj, post-attention add-RMSNorm, router, bucketing, grouped
    gate|up + SiLU*mul and grouped down x routing weight of ONE token in one persistent launch (loader run-ahead over the
    seams, granule hand-offs) — against the three launches it replaces (glm_gemm_bf16_router_tail,
    glm_grouped_gemm_gateup_silu, glm_grouped_gemm_down on the one-token GEMV body): EVERY stage bit for bit at its own
    scale (normalised row, residual, routing, bucketing, act rows, y rows).  New inputs and new weights' rows in use every
    repetition, unrelated launches in between (a stale granule of an earlier epoch, or a hand-off read early, would show);
    then the same launch replayed from a HIP graph with fresh inputs (the epoch lives in device memory)."""
    wo = dev(rnd(H, K, seed=411, scale=K ** -0.5))
    nw = dev(rnd(H, seed=412) * 0.1 + 1.0)
    gw = dev(rnd(E, H, seed=413, scale=H ** -0.5))
    bias = dev((torch.rand(E, generator=torch.Generator().manual_seed(414)) * 0.2 - 0.1).float())
    w13 = ops.padded_weight(2 * I, H, E + 1, dtype=BF, device="cuda")
    w2 = ops.padded_weight(H, I, E + 1, dtype=BF, device="cuda")
    gg = torch.Generator(device="cuda").manual_seed(415)   # GBs at Air dims: drawn on the device
    w13.normal_(0.0, H ** -0.5, generator=gg)
    w2.normal_(0.0, I ** -0.5, generator=gg)

    def three_launches(x, res):
        r = ops.gemm_add_rmsnorm_router_topk_align(x, wo, res, nw, 1e-5, gw, bias, k, 2.5, True)
        assert r is not None
        yn, ro, idx, tw, (offs, tok, scale, pos) = r
        act = ops.grouped_gemm_gateup_silu(yn.view(1, H), w13, offs, tok, k + 1, max_rows_per_group=1)
        y = ops.grouped_gemm_down(act, w2, offs, scale, E, max_rows_per_group=1)
        return yn, ro, idx, tw, (offs, tok, scale, pos), y, act

    def compare(a, b, what):
        names = ("y_norm", "residual_out", "topk_idx", "topk_w")
        for n, u, v in zip(names, a[:4], b[:4]):
            assert torch.equal(u, v), f"{what}: {n} differs"
        for n, u, v in zip(("offsets", "sorted_token", "sorted_scale", "pair_pos"), a[4], b[4]):
            assert torch.equal(u, v), f"{what}: {n} differs"
        assert torch.equal(a[6], b[6]), f"{what}: act rows differ ({int((a[6] != b[6]).sum())} elements)"
        assert torch.equal(a[5], b[5]), f"{what}: y rows differ ({int((a[5] != b[5]).sum())} elements)"

    with ops.option("t1_engine", 1):
        assert ops.can_decode_mlp_engine(dev(rnd(1, K, seed=1)), H, E, k, I)
    for rep in range(6):
        x, res = dev(rnd(1, K, seed=420 + rep)), dev(rnd(1, H, seed=430 + rep))
        ref = three_launches(x, res)
        got = ops.decode_mlp_engine(x, wo, res, nw, 1e-5, gw, bias, k, 2.5, w13, w2, want_act=True)
        assert got is not None, "the engine must cover this shape"
        ops.rmsnorm(dev(rnd(64, 4096, seed=rep)), dev(rnd(4096, seed=7)), 1e-5)
        torch.cuda.synchronize()
        ops.check_decode_engine()
        compare(ref, got, f"rep {rep}")
        _check_align(*got[4], got[2].cpu(), got[3].cpu(), E, True)
    # against the oracle: expert rows of the last repetition (routing taken from the product; GEMV fp32 order differs from the CPU's)
    yn, idx, tw = got[0].cpu(), got[2].cpu()[0], got[3].cpu()[0]
    for j in range(k):
        e = int(idx[j])
        ref_row = (O.mlp(yn.view(1, H), w13[e].cpu(), w2[e].cpu()).float() * tw[j].float()).to(BF)
        assert_close_bf16(got[5][j].cpu().view(1, H), ref_row, 2, 0.06, atol=2 ** -7, what=f"expert row {j} vs oracle")
    assert_close_bf16(got[5][k].cpu().view(1, H), O.mlp(yn.view(1, H), w13[E].cpu(), w2[E].cpu()), 2, 0.06, atol=2 ** -7,
                      what="shared row vs oracle")
    # HIP-graph replay: static inputs refilled between replays; every replay is a new epoch
    xs, rs = dev(rnd(1, K, seed=450)), dev(rnd(1, H, seed=451))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.decode_mlp_engine(xs, wo, rs, nw, 1e-5, gw, bias, k, 2.5, w13, w2, want_act=True)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = ops.decode_mlp_engine(xs, wo, rs, nw, 1e-5, gw, bias, k, 2.5, w13, w2, want_act=True)
    for rep in range(4):
        xs.copy_(rnd(1, K, seed=460 + rep)); rs.copy_(rnd(1, H, seed=470 + rep))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        ops.check_decode_engine()
        compare(three_launches(xs, rs), out, f"graph replay {rep}")
    st = next(iter(ops._engine_ws.values()))[:8].view(torch.int32)
    assert int(st[1]) == 0 and int(st[0]) > 0, "status clean, epoch advanced"


@pytest.mark.parametrize("M,N,K,want", [(32, 4096, 12288, 8), (5, 256, 1024, 3), (1, 64, 136, 2), (17, 512, 4096, 64)])
def test_gemm_partial_and_fused_add_rmsnorm(ops, M, N, K, want):
    """K-slice partial sums + the add-RMSNorm that reduces them == linear -> add_rms_forward of the oracle."""
    x = rnd(M, K, seed=171)
    w = rnd(N, K, seed=172, scale=K ** -0.5)
    res = rnd(M, N, seed=173)
    nw = rnd(N, seed=174) * 0.1 + 1.0
    ps = ops.gemm_bf16_partial(dev(x), dev(w), want)
    ns = ps.part.shape[0]
    assert 1 < ns <= want and ps.part.shape[1:] == (M, N)
    tot = ps.part[0].clone()
    for i in range(1, ns):  # the kernel's own order: slice 0 + slice 1 + ...
        tot += ps.part[i]
    lin = O.linear(x, w, None)
    # the summed partials round to the same bf16 values as the one-pass kernel, up to fp32 summation order
    assert_close_bf16(tot.to(BF), lin, 1, 0.05, atol=2 ** -9, what=f"partials {M}x{N}x{K}")
    y, ro = ops.add_rmsnorm_partial(ps, dev(res), dev(nw), 1e-5)
    # exact against the product's own two-kernel sequence whenever the rounded linear outputs agree
    y2, ro2 = ops.add_rmsnorm(tot.to(BF), dev(res), dev(nw), 1e-5)
    assert torch.equal(ro, ro2) and torch.equal(y, y2)
    ye, re_ = O.add_rms_forward(lin, res, nw, 1e-5)
    assert_close_bf16(ro, re_, 2, 0.05, atol=2 ** -8, what="residual")  # 1 ulp of the linear output + the add
    assert_close_bf16(y, ye, 2, 0.05, atol=2 ** -7, what="normed")


@pytest.mark.parametrize("T,H,E,k,partial", [(32, 4096, 64, 8, False), (32, 4096, 64, 8, True), (5, 512, 16, 4, False),
                                              (3, 8192, 128, 8, True), (130, 1024, 8, 2, False),
                                              # one token: the selecting wave buckets, no ticket / bucketing pass (r03)
                                              (1, 4096, 64, 8, False), (1, 4096, 64, 8, True), (1, 512, 16, 4, False),
                                              (1, 2048, 128, 8, False), (1, 1024, 72, 3, False)])
def test_fused_add_rmsnorm_router(ops, T, H, E, k, partial):
    """One launch == add_rmsnorm (or its partial-sum form) followed by router_topk, bit for bit."""
    g = torch.Generator().manual_seed(T + H + E)
    res = rnd(T, H, seed=181)
    nw = rnd(H, seed=182) * 0.1 + 1.0
    gw = rnd(E, H, seed=183, scale=H ** -0.5)
    bias = (torch.rand(E, generator=g) * 0.2 - 0.1).float()
    if partial:
        K = 1024
        ps = ops.gemm_bf16_partial(dev(rnd(T, K, seed=184)), dev(rnd(H, K, seed=185, scale=K ** -0.5)), 4)
        y0, r0 = ops.add_rmsnorm_partial(ps, dev(res), dev(nw), 1e-5)
        x = ps
    else:
        x = dev(rnd(T, H, seed=186))
        y0, r0 = ops.add_rmsnorm(x, dev(res), dev(nw), 1e-5)
    i0, w0 = ops.router_topk(y0, dev(gw), dev(bias), k, 2.5)
    y1, r1, i1, w1 = ops.add_rmsnorm_router_topk(x, dev(res), dev(nw), 1e-5, dev(gw), dev(bias), k, 2.5)
    assert torch.equal(y0, y1) and torch.equal(r0, r1)
    assert torch.equal(i0, i1) and torch.equal(w0, w1)
    # ... and with the token -> expert bucketing done by the launch's last workgroup (agent-scope hand-off): launched
    # repeatedly on the same ticket, between unrelated launches that keep the other CUs busy
    for rep in range(4):
        for shared in (True, False):
            y2, r2, i2, w2, al = ops.add_rmsnorm_router_topk(x, dev(res), dev(nw), 1e-5, dev(gw), dev(bias), k, 2.5,
                                                             align_with_shared=shared)
            ops.rmsnorm(dev(rnd(64, 4096, seed=rep)), dev(rnd(4096, seed=7)), 1e-5)
            assert torch.equal(y0, y2) and torch.equal(r0, r2) and torch.equal(i0, i2) and torch.equal(w0, w2)
            _check_align(*al, i2.cpu(), w2.cpu(), E, shared)
    head = ops._router_workspace(y1.device, T, E)[:262208].view(torch.int32)
    assert int(head.abs().sum()) == 0, "every arrival counter must be back at zero"


def test_gemm_bf16_strided_input(ops):
    M, N, K = 12, 256, 384
    big = rnd(M, K + 64, seed=74)
    w = rnd(N, K, seed=75, scale=K ** -0.5)
    out = ops.gemm_bf16(dev(big)[:, :K], dev(w), None)
    assert_close_bf16(out, O.linear(big[:, :K], w), 1, 0.05, what="gemm strided")


def test_error_reporting(ops):
    x = rnd(4, 100).cuda()  # K % 8 != 0 is unsupported by the streaming kernel
    w = rnd(8, 100).cuda()
    with pytest.raises(RuntimeError, match="gemm"):
        ops.gemm_bf16(x, w, None)
    with pytest.raises(RuntimeError, match="GPU tensors"):
        ops.rmsnorm(rnd(2, 8), rnd(8), 1e-5)


# ------------------------------------------------------------------------------------------
# prefill regime: LDS-tiled MFMA GEMM (avg rows/group >= 96) and varlen causal attention
@pytest.mark.parametrize("M,N,K,bias", [(96, 64, 64, False), (128, 200, 4096, True), (200, 4096, 1408, False),
                                        (1000, 384, 352, True), (130, 14336 // 4, 512, True)])
def test_gemm_bf16_large_m(ops, M, N, K, bias):
    x = rnd(M, K, seed=81)
    w = rnd(N, K, seed=82, scale=K ** -0.5)
    b = rnd(N, seed=83) if bias else None
    out = ops.gemm_bf16(dev(x), dev(w), dev(b))
    assert_close_bf16(out, O.linear(x, w, b), 1, 0.05, atol=2 ** -9, what=f"gemm(large) {M}x{N}x{K}")


@pytest.mark.parametrize("K", [64, 128, 192, 320, 512])
def test_gemm_large_m_quadrant_loop_equals_plain_loop(ops, K):
    """The default large-M loop (256x256x64 tiles on two LDS buffers, operand slots refilled as soon as their own last
    fragment read retired; csrc/gemm_largem.hip LM_PH8) against variant 0 of the same kernel (4-stage ring, 256x256x32),
    bit for bit, at 1, 2, 3 and 5 K tiles (prologue / steady state / drain of the two-buffer schedule), for the plain,
    gate|up and down epilogues, over ragged groups (one empty, one with a partial last row block)."""
    lib = ops.load_library()
    H, I = K, 192
    counts = [300, 0, 513, 256, 211]                      # average 256 rows per group: the 256-row tile configuration
    E, R = len(counts) - 1, sum(counts)
    offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32)
    g = torch.Generator().manual_seed(K)
    tok = torch.randint(0, 400, (R,), generator=g, dtype=torch.int32)
    x = rnd(400, H, seed=70 + K)
    w13 = rnd(E + 1, 2 * I, H, seed=71, scale=H ** -0.5)
    scale = (torch.rand(R, generator=g) + 0.1).to(BF)
    act_in = rnd(R, K, seed=72)
    w2 = rnd(E + 1, 136, K, seed=73, scale=K ** -0.5)      # N = 136: a partial column tile
    xp, wp = rnd(700, K, seed=74), rnd(264, K, seed=75, scale=K ** -0.5)
    bias = rnd(264, seed=76)

    def run():
        a = ops.grouped_gemm_gateup_silu(dev(x), dev(w13), dev(offs), dev(tok), R)
        y = ops.grouped_gemm_down(dev(act_in), dev(w2), dev(offs), dev(scale), E)
        z = ops.gemm_bf16(dev(xp), dev(wp), dev(bias))
        torch.cuda.synchronize()
        return a.cpu(), y.cpu(), z.cpu()

    with largem_variant(0):
        ref = run()
    got = run()
    for name, a, b in zip(("gate|up", "down", "plain"), got, ref):
        assert torch.equal(a, b), f"{name}: quadrant loop differs from variant 0 at K={K}"
    # and variant 0 itself against the oracle on two groups
    W13 = w13
    for e in (0, 4):
        rows = slice(int(offs[e]), int(offs[e + 1]))
        a_ref = O.silu_and_mul(O.linear(x[tok[rows].long()], W13[e]))
        assert_close_bf16(got[0][rows], a_ref, 3, 0.06, atol=1e-3, what=f"act K={K} group {e}")
    assert_close_bf16(got[2], O.linear(xp, wp, bias), 1, 0.05, atol=2 ** -9, what=f"plain K={K}")


@pytest.mark.parametrize("T,nh,nkv,K,with_cache,need_kv", [(1024, 12, 2, 512, True, True), (300, 4, 1, 256, True, False),
                                                          (777, 6, 2, 192, False, True), (4096, 96, 8, 4096, True, True)])
def test_qkv_gemm_with_rope_store_epilogue(ops, T, nh, nkv, K, with_cache, need_kv):
    """r04: glm_gemm_qkv_rope_store (QKV projection whose epilogue splits, rotates q / k and scatters K / V into the paged
    cache) == glm_gemm_bf16 followed by glm_qkv_split_norm_rope_store, bit for bit — q, k, v and every cache row, rows with
    slot -1 untouched — and both against the oracle (linear.py:74-75, rotary_embedding.py:39-55, attention.py:10-42).  The
    last case is the Air shape (H = 4096, 96 / 8 heads); ragged T exercises partial row blocks."""
    hd, rd = 128, 64
    N = (nh + 2 * nkv) * hd
    x = rnd(T, K, seed=901)
    w = rnd(N, K, seed=902, scale=K ** -0.5)
    b = rnd(N, seed=903)
    g = torch.Generator().manual_seed(904)
    pos = torch.randint(0, 2048, (T,), generator=g, dtype=torch.int64)
    cs = O.build_cos_sin_cache(rd, 2048, 10000.0).reshape(2048, rd).contiguous()
    nblk = (T + 255) // 256 + 2
    slots = torch.randperm(nblk * 256, generator=g)[:T].to(torch.int32)
    slots[::7] = -1
    kc0 = rnd(nblk, 256, nkv, hd, seed=905)
    vc0 = rnd(nblk, 256, nkv, hd, seed=906)

    def two_launches():
        kc, vc = dev(kc0), dev(vc0)
        qkv = ops.gemm_bf16(dev(x), dev(w), dev(b))
        q, k, v = ops.qkv_split_norm_rope_store(qkv, nh, nkv, hd, dev(pos), dev(cs), rd, None, None, 1e-5,
                                                kc if with_cache else None, vc if with_cache else None,
                                                dev(slots) if with_cache else None, need_kv)
        return q, k, v, kc, vc

    def one_launch():
        kc, vc = dev(kc0), dev(vc0)
        r = ops.gemm_qkv_rope_store(dev(x), dev(w), dev(b), nh, nkv, hd, dev(pos), dev(cs), rd,
                                    kc if with_cache else None, vc if with_cache else None,
                                    dev(slots) if with_cache else None, need_kv)
        assert r is not None, "the fused launch must cover this shape"
        return (*r, kc, vc)

    q0, k0, v0, kc_a, vc_a = two_launches()
    for rep in range(2):
        q1, k1, v1, kc_b, vc_b = one_launch()
        torch.cuda.synchronize()
        assert torch.equal(q0, q1), "q differs"
        if need_kv:
            assert torch.equal(k0, k1) and torch.equal(v0, v1), "k / v differ"
        else:
            assert k1 is None and v1 is None
        assert torch.equal(kc_a, kc_b) and torch.equal(vc_a, vc_b), "cache rows differ"
    if with_cache:
        skipped = (slots < 0)
        keep = torch.ones(nblk * 256, dtype=torch.bool)
        keep[slots[~skipped].long()] = False
        assert torch.equal(kc_b.cpu().view(-1, nkv * hd)[keep], kc0.view(-1, nkv * hd)[keep]), "untouched cache rows changed"
    # oracle on a sample of rows (the GEMM's fp32 summation order differs from the CPU's by single ulps)
    rows = torch.arange(0, T, max(1, T // 64))
    qkv_ref = O.linear(x[rows], w, b)
    qr, kr, vr = qkv_ref.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
    qr, kr = O.rope_neox(pos[rows], qr.reshape(-1, nh, hd), kr.reshape(-1, nkv, hd), cs.reshape(2048, 1, rd), rd)
    assert_close_bf16(q1.cpu()[rows], qr, 2, 0.06, atol=2 ** -6, what="q vs oracle")
    if need_kv:
        assert_close_bf16(k1.cpu()[rows], kr, 2, 0.06, atol=2 ** -6, what="k vs oracle")
        assert_close_bf16(v1.cpu()[rows], vr.reshape(-1, nkv, hd), 1, 0.05, atol=2 ** -9, what="v vs oracle")
    # below the tiled kernel's row count the caller keeps the two launches
    assert ops.gemm_qkv_rope_store(dev(x[:32]), dev(w), dev(b), nh, nkv, hd, dev(pos[:32]), dev(cs), rd, None, None, None, True) is None
    # a loop variant that parks epilogue stores for the next tile (LM_PARK) has no such epilogue: refused, the caller keeps
    # the two launches (it used to flush never-written LDS through q's rows)
    with ops.option("largem_variant", 348931):
        assert ops.gemm_qkv_rope_store(dev(x), dev(w), dev(b), nh, nkv, hd, dev(pos), dev(cs), rd, None, None, None, need_kv) is None
    # a slot_mapping that does not match the rows is an error, never a silently skipped cache store
    if with_cache:
        with pytest.raises(ValueError, match="slot_mapping"):
            ops.gemm_qkv_rope_store(dev(x), dev(w), dev(b), nh, nkv, hd, dev(pos), dev(cs), rd, dev(kc0), dev(vc0),
                                    dev(slots[: T - 1]), need_kv)


@pytest.mark.parametrize("K,counts", [(256, [2048, 1500, 3000, 1000, 6000]), (512, [5000, 0, 257, 4096, 3000]),
                                      (320, [4000, 4000, 4000, 4000]), (128, [3000, 5000, 2000])])
def test_gemm_large_m_several_tiles_per_workgroup(ops, K, counts):
    """r05: ~500 output tiles on 256 persistent workgroups — the tile switch of the product loop (tile table, ring fill in
    front of the epilogue, store-count wait at the top of the next tile) runs once or twice in every workgroup, with even
    (4, 8) and odd (5) K-tile counts and a two-tile K loop.  Bit for bit against variant 0, three launches each (a late or
    misdirected DMA piece would show as a rare wrong tile); one group also against the oracle (mlp.py:49-54).  In a STUDY
    build variant 611075 adds LM_CONT (the next tile's first K tiles issued from the current tile's last two: measured
    2-3 % slower, csrc/gemm_largem.hip); the product library maps the number to variant 0."""
    I, N2 = 1024, 2048
    E, R = len(counts) - 1, sum(counts)
    offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32)
    g = torch.Generator().manual_seed(K)
    T = 3000
    tok = torch.randint(0, T, (R,), generator=g, dtype=torch.int32)
    x = rnd(T, K, seed=K + 1)
    w13 = rnd(E + 1, 2 * I, K, seed=K + 2, scale=K ** -0.5)
    scale = (torch.rand(R, generator=g) + 0.1).to(BF)
    act_in = rnd(R, K, seed=K + 3)
    w2 = rnd(E + 1, N2, K, seed=K + 4, scale=K ** -0.5)
    xd, w13d, offd, tokd, actd, w2d, scd = dev(x), dev(w13), dev(offs), dev(tok), dev(act_in), dev(w2), dev(scale)

    def run():
        a = ops.grouped_gemm_gateup_silu(xd, w13d, offd, tokd, R)
        y = ops.grouped_gemm_down(actd, w2d, offd, scd, E)
        torch.cuda.synchronize()
        return a, y

    with largem_variant(0):
        ref = run()
    with largem_variant(86787):
        base = run()
    assert torch.equal(base[0], ref[0]) and torch.equal(base[1], ref[1])
    for v in (611075, -1):
        with largem_variant(v):
            for _ in range(3):
                got = run()
                assert torch.equal(got[0], ref[0]), f"gate|up: variant {v} differs from variant 0 (K={K} counts={counts})"
                assert torch.equal(got[1], ref[1]), f"down: variant {v} differs from variant 0 (K={K} counts={counts})"
    e = E
    rows = slice(int(offs[e]), int(offs[e]) + min(300, counts[e]))
    a_ref = O.silu_and_mul(O.linear(x[tok[rows].long()], w13[e]))
    assert_close_bf16(got[0][rows].cpu(), a_ref, 3, 0.06, atol=1e-3, what=f"act K={K} group {e}")


@pytest.mark.parametrize("seed", range(6))
def test_gemm_large_m_quadrant_loop_random_shapes(ops, seed):
    """Random ragged groupings (empty groups, single-row groups, partial row blocks and column tiles), random K in whole
    64-wide tiles: default loop == variant 0, bit for bit, three launches each (a late DMA would show as a rare wrong tile)."""
    lib = ops.load_library()
    g = torch.Generator().manual_seed(1000 + seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    K = 64 * ri(1, 9)
    I = 8 * ri(3, 40)
    N2 = 8 * ri(5, 70)
    E = ri(1, 6)
    counts = [ri(0, 700) if ri(0, 4) else 0 for _ in range(E + 1)]
    counts[ri(0, E)] += 256 * (E + 1)           # keeps the average at >= 256 rows per group: the 256-row configuration
    R = sum(counts)
    offs = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32)
    T = ri(50, 900)
    tok = torch.randint(0, T, (R,), generator=g, dtype=torch.int32)
    x = rnd(T, K, seed=seed + 1)
    w13 = rnd(E + 1, 2 * I, K, seed=seed + 2, scale=K ** -0.5)
    scale = (torch.rand(R, generator=g) + 0.1).to(BF)
    act_in = rnd(R, K, seed=seed + 3)
    w2 = rnd(E + 1, N2, K, seed=seed + 4, scale=K ** -0.5)

    def run():
        a = ops.grouped_gemm_gateup_silu(dev(x), dev(w13), dev(offs), dev(tok), R)
        y = ops.grouped_gemm_down(dev(act_in), dev(w2), dev(offs), dev(scale), E)
        torch.cuda.synchronize()
        return a.cpu(), y.cpu()

    with largem_variant(0):
        ref = run()
    for _ in range(3):
        got = run()
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), \
            f"quadrant loop differs from variant 0 (K={K} I={I} N={N2} counts={counts})"
    e = max(range(E + 1), key=lambda i: counts[i])
    rows = slice(int(offs[e]), int(offs[e + 1]))
    a_ref = O.silu_and_mul(O.linear(x[tok[rows].long()], w13[e]))
    assert_close_bf16(ref[0][rows], a_ref, 3, 0.06, atol=1e-3, what=f"act, random shapes seed {seed}")


@pytest.mark.parametrize("T,E,k,H,I", [(600, 4, 2, 256, 128), (1024, 8, 2, 128, 352), (300, 2, 1, 512, 64)])
def test_moe_block_large_t(ops, T, E, k, H, I):
    x = rnd(T, H, seed=91)
    w13, w2 = rnd(E, 2 * I, H, seed=92, scale=H ** -0.5), rnd(E, H, I, seed=93, scale=I ** -0.5)
    sw13, sw2 = rnd(2 * I, H, seed=94, scale=H ** -0.5), rnd(H, I, seed=95, scale=I ** -0.5)
    g = torch.Generator().manual_seed(96)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    ref = O.moe(x, None, None, k, 1.0, w13, w2, sw13, sw2, topk=(idx, w))
    out, (_, _, offs, tok, scale, pos, act, y) = _moe_hip(ops, x, None, None, k, 1.0, w13, w2, sw13, sw2, topk=(idx, w))
    offs, tok = offs.cpu(), tok.cpu()
    W13 = torch.cat([w13, sw13.unsqueeze(0)])
    for e in range(E + 1):
        rows = slice(int(offs[e]), int(offs[e + 1]))
        if rows.start == rows.stop:
            continue
        a_ref = O.silu_and_mul(O.linear(x[tok[rows].long()], W13[e]))
        # gate, up and bf16(silu) can each flip one ulp (fp32 summation order): <= 3 ulp stacked
        assert_close_bf16(act[rows], a_ref, 3, 0.06, atol=1e-3, what=f"act(large) expert {e}")
    assert_mixed(out, ref, 2 ** -6, 2 ** -6 * float(ref.float().abs().max()), "moe block (large T)")


@pytest.mark.parametrize("lens_q,lens_k,nh,nkv,paged", [
    ([21, 16], [21, 16], 4, 2, False), ([64], [64], 2, 1, False), ([65, 1, 200], [65, 1, 200], 12, 2, False),
    ([300], [300], 24, 2, False), ([5, 40], [300, 296], 4, 2, True), ([130], [513], 8, 8, True),
    # (sequences x kv heads) divisible by 8: the XCD-grouped workgroup order; several query blocks per sequence
    ([300, 17, 513, 64], [300, 17, 513, 64], 16, 4, False), ([700, 260], [700, 260], 8, 4, False),
    ([100, 50], [612, 306], 16, 4, True), ([1025], [1025], 12, 1, False),
    # r05: contiguous K/V with fewer queries than keys (a prefix already in K/V) and with MORE queries than keys (bottom-right
    # aligned mask: the first rows see nothing and come out as zeros); a two-block sequence whose last tile is partial next
    # to one whose tiles are all full (the predicate-free loop's range ends one tile earlier in the first)
    ([5, 40], [300, 296], 4, 2, False), ([200, 70], [40, 70], 4, 2, False), ([250, 256], [250, 256], 8, 2, False),
    ([0, 33], [0, 33], 4, 2, False)])
def test_varlen_attn_prefill(ops, lens_q, lens_k, nh, nkv, paged):
    g = torch.Generator().manual_seed(sum(lens_k) + nh)
    hd, bs = 128, 256
    Tq = sum(lens_q)
    q = torch.randn(Tq, nh, hd, generator=g).to(BF)
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32)
    scale = hd ** -0.5
    if paged:
        _, kc, vc, _, bt = _paged_case(lens_k, nh, nkv, seed=5)
        ref = O.attn_prefill_varlen(q, kc, vc, cu_q, cu_k, scale, bt)
        out = ops.varlen_attn_prefill(dev(q), dev(kc), dev(vc), dev(cu_q), dev(cu_k), max(lens_q), scale,
                                      dev(bt), bs)
    else:
        Tk = sum(lens_k)
        k = torch.randn(Tk, nkv, hd, generator=g).to(BF)
        v = torch.randn(Tk, nkv, hd, generator=g).to(BF)
        ref = O.attn_prefill_varlen(q, k, v, cu_q, cu_k, scale)
        out = ops.varlen_attn_prefill(dev(q), dev(k), dev(v), dev(cu_q), dev(cu_k), max(lens_q), scale)
    assert torch.isfinite(out.float()).all()
    assert_mixed(out, ref, 1.6e-2, 4e-3, f"prefill attn q={lens_q} k={lens_k}")


def test_study_kernel_options_are_refused_by_the_product_library(ops, opt_guard):
    """r05 housekeeping: the measured-and-lost prefill attention kernels (32x32x16 MFMAs, 64 query rows per wave, persistent
    workgroups) live in the
    study library only (build.py --study; tests/study_kernels.py runs there).  The product library refuses their options
    loudly instead of shipping them."""
    q = dev(rnd(33, 8, 128, seed=1))
    k = dev(rnd(33, 2, 128, seed=2))
    cu = dev(torch.tensor([0, 33], dtype=torch.int32))
    for name in ("fa_mfma32", "fa_rows64", "fa_persist"):
        with ops.option(name, 1):
            with pytest.raises(RuntimeError, match="study"):
                ops.varlen_attn_prefill(q, k, k, cu, cu, 33, 128 ** -0.5)
    ops.varlen_attn_prefill(q, k, k, cu, cu, 33, 128 ** -0.5)


def test_padded_weight_pitch(ops):
    """Row-padded weight storage behind [out, in] / [G, 2I, H] views gives identical results."""
    M, N, K = 32, 512, 4096
    x = rnd(M, K, seed=101)
    w = rnd(N, K, seed=102, scale=K ** -0.5)
    wp = ops.padded_weight(N, K, dtype=BF, device="cuda")
    assert wp.shape == (N, K) and wp.stride(0) == K + 64
    wp.copy_(w)
    assert torch.equal(ops.gemm_bf16(dev(x), wp, None), ops.gemm_bf16(dev(x), dev(w), None))
    big = rnd(200, K, seed=103)
    assert torch.equal(ops.gemm_bf16(dev(big), wp, None), ops.gemm_bf16(dev(big), dev(w), None))
    # grouped: padded stacks
    T, E, k, H, I = 16, 4, 2, 2048, 64
    xs = rnd(T, H, seed=104)
    w13 = rnd(E + 1, 2 * I, H, seed=105, scale=H ** -0.5)
    w13p = ops.padded_weight(2 * I, H, E + 1, dtype=BF, device="cuda")
    assert w13p.stride(1) == H + 64
    w13p.copy_(w13)
    g = torch.Generator().manual_seed(106)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    wts = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    offs, tok, scale, pos = ops.moe_align(dev(idx), dev(wts), E, True)
    R = T * k + T
    a1 = ops.grouped_gemm_gateup_silu(dev(xs), w13p, offs, tok, R)
    a2 = ops.grouped_gemm_gateup_silu(dev(xs), dev(w13), offs, tok, R)
    assert torch.equal(a1, a2)


# ------------------------------------------------------------------------------------------
# N4: fp8 (OCP e4m3) weight-only quantised experts
@pytest.mark.parametrize("T,E,k,H,I", [(32, 8, 2, 512, 128), (5, 4, 2, 4096, 192), (70, 4, 2, 256, 352),
                                       # many rows per group: exact fp8 -> bf16 expansion + the tiled MFMA kernel
                                       (600, 4, 2, 256, 128), (1024, 2, 1, 512, 352)])
def test_moe_block_fp8_weights(ops, T, E, k, H, I):
    x = rnd(T, H, seed=111)
    w13, w2 = rnd(E + 1, 2 * I, H, seed=112, scale=H ** -0.5), rnd(E + 1, H, I, seed=113, scale=I ** -0.5)
    g = torch.Generator().manual_seed(114)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    # quantise once (product side) and give the oracle the same bytes and scales; the quantiser
    # itself is checked against the oracle's by its reconstruction error
    q13d, s13d = ops.quantize_fp8_rowwise(dev(w13))
    q2d, s2d = ops.quantize_fp8_rowwise(dev(w2))
    q13, s13 = q13d.cpu().contiguous().view(torch.float8_e4m3fn), s13d.cpu()
    q2, s2 = q2d.cpu().contiguous().view(torch.float8_e4m3fn), s2d.cpu()
    oq, os_ = O.quantize_fp8_rowwise(w13)
    assert torch.allclose(os_, s13, rtol=1e-6)
    # the GPU's float→e4m3 conversion and torch-CPU's differ on a small fraction of elements
    # (ties / fp8-subnormal range); the bytes the kernels see are the product's either way
    assert float((oq.float() != q13.float()).float().mean()) < 0.03, "quantisers disagree beyond rounding cases"
    deq = q13.float() * s13.unsqueeze(-1)
    assert float((deq - w13.float()).abs().max() / w13.float().abs().max()) < 0.07  # e4m3: 3 mantissa bits
    if H % 4096 == 0:
        assert q13d.stride(1) == H + 128  # padded fp8 row pitch
    offs, tok, scale, pos = ops.moe_align(dev(idx), dev(w), E, True)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu_fp8w(dev(x), q13d, s13d, offs, tok, R)
    y = ops.grouped_gemm_down_fp8w(act, q2d, s2d, offs, scale, E)
    out = ops.moe_combine(y, pos, T, k, T * k)
    offs_c, tok_c = offs.cpu(), tok.cpu()
    for e in range(E + 1):
        rows = slice(int(offs_c[e]), int(offs_c[e + 1]))
        if rows.start == rows.stop:
            continue
        a_ref = O.silu_and_mul(O.linear_fp8w(x[tok_c[rows].long()], q13[e], s13[e]))
        assert_close_bf16(act[rows], a_ref, 3, 0.06, atol=1e-3, what=f"fp8 act expert {e}")
    ref = O.moe_fp8w(x, k, idx, w, q13[:E], s13[:E], q2[:E], s2[:E], q13[E], s13[E], q2[E], s2[E])
    assert_mixed(out, ref, 2 ** -6, 2 ** -6 * float(ref.float().abs().max()), "moe block fp8 weights")
    # and the quantisation error itself is what fp8 e4m3 predicts (a few percent), not a bug
    ref_bf16 = O.moe(x, None, None, k, 1.0, w13[:E], w2[:E], w13[E], w2[E], topk=(idx, w))
    rel = float((ref.float() - ref_bf16.float()).norm() / ref_bf16.float().norm())
    assert rel < 0.08, f"fp8 quantisation error {rel:.3f}"


@pytest.mark.parametrize("T,E,k,H,I", [(600, 4, 2, 256, 128),     # 256-row blocks: the quadrant loop with the fp8 operand
                                       (300, 8, 2, 512, 192),     # ~100 rows per group: 128-row blocks, plain loop
                                       (2100, 6, 3, 1024, 1408),  # Air's I (22 K tiles of the down projection), ragged groups
                                       (700, 5, 1, 4096, 64)])    # one expert never chosen (an empty group)
def test_fp8_weights_native_operand_equals_expansion(ops, T, E, k, H, I, opt_guard):
    """Many rows per group: the tiled kernel fed with the fp8 bytes (DMA-staged as fp8, converted between the fragment read
    and the MFMA) == the exact fp8 -> bf16 expansion pass followed by the bf16 kernel, bit for bit, both projections."""
    x = rnd(T, H, seed=121)
    w13, w2 = rnd(E + 1, 2 * I, H, seed=122, scale=H ** -0.5), rnd(E + 1, H, I, seed=123, scale=I ** -0.5)
    g = torch.Generator().manual_seed(124)
    pool = E - 1 if E == 5 else E
    idx = torch.stack([torch.randperm(pool, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    q13, s13 = ops.quantize_fp8_rowwise(dev(w13))
    q2, s2 = ops.quantize_fp8_rowwise(dev(w2))
    offs, tok, scale, pos = ops.moe_align(dev(idx), dev(w), E, True)
    R = T * k + T
    opt_guard("fp8w_native", 0)
    a0 = ops.grouped_gemm_gateup_silu_fp8w(dev(x), q13, s13, offs, tok, R)
    y0 = ops.grouped_gemm_down_fp8w(a0, q2, s2, offs, scale, E)
    opt_guard("fp8w_native", 1)
    a1 = ops.grouped_gemm_gateup_silu_fp8w(dev(x), q13, s13, offs, tok, R)
    y1 = ops.grouped_gemm_down_fp8w(a0, q2, s2, offs, scale, E)
    assert torch.equal(a0, a1), "gate|up: native fp8 operand differs from the expansion path"
    assert torch.equal(y0, y1), "down: native fp8 operand differs from the expansion path"
    assert float(a1.float().abs().max()) > 0


# ------------------------------------------------------------------------------------------
# N3: expert-parallel dispatch / combine pieces
def test_row_gather_scatter(ops):
    for n_src, cols in ((37, 4096), (5, 100), (64, 1408)):
        src = rnd(n_src, cols, seed=n_src)
        g = torch.Generator().manual_seed(cols)
        idx = torch.randint(0, n_src, (91,), generator=g).to(torch.int32)
        assert torch.equal(ops.gather_rows(dev(src), dev(idx)).cpu(), src[idx.long()])
        perm = torch.randperm(n_src, generator=g).to(torch.int32)
        out = ops.scatter_rows(dev(src), dev(perm)).cpu()
        exp = torch.zeros_like(src)
        exp[perm.long()] = src
        assert torch.equal(out, exp)
    idx = torch.tensor([3, -1, 0], dtype=torch.int32)  # -1 skips
    out = ops.scatter_rows(dev(rnd(3, 64, seed=1)), dev(idx), 5).cpu()
    assert float(out[1].float().abs().max()) == 0 and float(out[2].float().abs().max()) == 0


@pytest.mark.parametrize("T,k,H", [(9, 4, 256), (32, 8, 4096), (3, 2, 100)])
def test_moe_combine_scaled_equals_scaled_rows_combine(ops, T, k, H):
    """Routing weight applied at combine time (expert-parallel path) == weight applied by the down projection's
    epilogue followed by the plain combine, bit for bit; shared rows from their own buffer."""
    P = T * k
    y = rnd(P, H, seed=201, scale=2.0)
    sc = (torch.rand(P, generator=torch.Generator().manual_seed(202)) + 0.1).to(BF)
    sh = rnd(T, H, seed=203)
    pos = torch.randperm(P, generator=torch.Generator().manual_seed(204)).view(T, k).to(torch.int32)
    got = ops.moe_combine_scaled(dev(y), dev(pos.flatten().contiguous()), dev(sc), dev(sh), T, k).cpu()
    ys = torch.cat([y * sc.unsqueeze(-1), sh])  # bf16 multiply = bf16(bf16(acc) * w)
    exp = ops.moe_combine(dev(ys), dev(pos.flatten().contiguous()), T, k, P).cpu()
    assert torch.equal(got, exp)


@pytest.mark.parametrize("ep,T,E,k,H,I", [(2, 32, 8, 2, 512, 128), (4, 37, 16, 4, 256, 192), (8, 5, 64, 8, 4096, 1408),
                                          (2, 1, 4, 2, 256, 64)])
def test_expert_parallel_emulated_equals_single_gpu(ops, ep, T, E, k, H, I):
    """EP = ep logical ranks on ONE device (same code, collectives done by copies) == the TP = 1 kernels' result: a
    row's value does not depend on which rank's grouped GEMM computed it nor on its position in the group — bit for
    bit whenever both sides run the same kernel instantiation; a rank that holds few rows may get another K-split
    configuration of the streaming GEMM (gemm_smallm.hip dispatches on the total row count), i.e. another fp32
    summation order: then single bf16 ulps."""
    from nanovllm.distributed import expert_parallel as EP
    x = rnd(T, H, seed=301)
    w13, w2 = rnd(E + 1, 2 * I, H, seed=302, scale=H ** -0.5), rnd(E + 1, H, I, seed=303, scale=I ** -0.5)
    g = torch.Generator().manual_seed(304)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    ref_dev, _ = _moe_hip(ops, x, None, None, k, 1.0, w13[:E], w2[:E], w13[E], w2[E], topk=(idx, w))
    W13, W2, xd = dev(w13), dev(w2), dev(x)
    El = E // ep

    def shared(hh):
        a = ops.grouped_gemm_gateup_silu(hh, W13[E:E + 1], None, None, hh.shape[0])
        return ops.grouped_gemm_down(a, W2[E:E + 1], None, None, 0)
    both = []
    for form in ("forward_fixed", "forward_varlen"):  # fixed-capacity slots (graph-capturable, the default) / exact splits
        gens = [getattr(EP.EPExperts(ops, ep, r, E, k, W13[r * El:(r + 1) * El], W2[r * El:(r + 1) * El], shared), form)
                (xd, dev(idx), dev(w)) for r in range(ep)]
        outs = EP.run_emulated(gens)
        for o in outs:
            assert torch.equal(o, outs[0])
        both.append(outs[0])
    assert torch.equal(both[0], both[1]), "fixed-capacity dispatch differs from the variable-split dispatch"
    if not torch.equal(outs[0], ref_dev):
        # sums of 8 + 1 rows that cancel to ~0 are compared absolutely (one ulp of a 0.5-sized row is 2^-9 .. 2^-8)
        assert_close_bf16(outs[0], ref_dev, 2, 0.05, atol=2 ** -5, what="expert-parallel vs single-GPU kernels")
    ref = O.moe(x, None, None, k, 1.0, w13[:E], w2[:E], w13[E], w2[E], topk=(idx, w))
    assert_mixed(outs[0], ref, 2 ** -6, 2 ** -6 * float(ref.float().abs().max()), "expert-parallel vs oracle")


@pytest.mark.parametrize("ep,T,E,k,H,I", [(2, 32, 8, 2, 512, 128), (4, 37, 16, 4, 256, 192), (8, 5, 64, 8, 4096, 1408)])
def test_expert_parallel_fp8_experts_emulated(ops, ep, T, E, k, H, I):
    """r04 (missing #4): fp8 e4m3 weight-only experts UNDER expert parallelism — ep logical ranks on one device,
    each streaming its own fp8 expert stack (and the replicated shared expert from its fp8 rows), both dispatch forms —
    against the single-GPU fp8 MoE block (same quantised weights; single bf16 ulps where a rank's row count selects another
    K-split of the streaming GEMM) and against the oracle on the DEQUANTISED weights (the N4 bar)."""
    from nanovllm.distributed import expert_parallel as EP
    x = rnd(T, H, seed=311)
    w13, w2 = rnd(E + 1, 2 * I, H, seed=312, scale=H ** -0.5), rnd(E + 1, H, I, seed=313, scale=I ** -0.5)
    g = torch.Generator().manual_seed(314)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    xd = dev(x)
    w13_q, w13_s = ops.quantize_fp8_rowwise(dev(w13))
    w2_q, w2_s = ops.quantize_fp8_rowwise(dev(w2))
    # single-GPU fp8 block
    offs, tok, scale, pos = ops.moe_align(dev(idx), dev(w), E, True)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu_fp8w(xd, w13_q, w13_s, offs, tok, R)
    y = ops.grouped_gemm_down_fp8w(act, w2_q, w2_s, offs, scale, E)
    ref_dev = ops.moe_combine(y, pos, T, k, T * k)
    El = E // ep

    def shared(hh):
        a = ops.grouped_gemm_gateup_silu_fp8w(hh, w13_q[E:E + 1], w13_s[E:E + 1], None, None, hh.shape[0])
        return ops.grouped_gemm_down_fp8w(a, w2_q[E:E + 1], w2_s[E:E + 1], None, None, 0)
    both = []
    for form in ("forward_fixed", "forward_varlen"):
        gens = [getattr(EP.EPExperts(ops, ep, r, E, k, None, None, shared,
                                     fp8=(w13_q[r * El:(r + 1) * El], w13_s[r * El:(r + 1) * El],
                                          w2_q[r * El:(r + 1) * El], w2_s[r * El:(r + 1) * El])), form)
                (xd, dev(idx), dev(w)) for r in range(ep)]
        outs = EP.run_emulated(gens)
        for o in outs:
            assert torch.equal(o, outs[0])
        both.append(outs[0])
    if not torch.equal(both[0], both[1]):
        # the fp8 launchers have no row-count hint: the capacity-sized row array of the fixed form may select another K-split
        # of the streaming GEMM than the exact rows of the variable form (another fp32 summation order)
        assert_close_bf16(both[0], both[1], 2, 0.05, atol=2 ** -5, what="fp8 experts: fixed-capacity vs variable-split dispatch")
    if not torch.equal(outs[0], ref_dev):
        assert_close_bf16(outs[0], ref_dev, 2, 0.05, atol=2 ** -5, what="fp8 experts, expert-parallel vs single-GPU kernels")
    # oracle on the product's own fp8 bytes and scales (what the fp8 block is specified against: weight-only, bf16 arithmetic)
    q13, s13 = w13_q.cpu().contiguous().view(torch.float8_e4m3fn), w13_s.cpu()
    q2, s2 = w2_q.cpu().contiguous().view(torch.float8_e4m3fn), w2_s.cpu()
    ref = O.moe_fp8w(x, k, idx, w, q13[:E], s13[:E], q2[:E], s2[:E], q13[E], s13[E], q2[E], s2[E])
    assert_mixed(outs[0], ref, 2 ** -6, 2 ** -6 * float(ref.float().abs().max()), "fp8 experts, expert-parallel vs oracle")


@pytest.mark.parametrize("T,E,k,ep", [(256, 64, 8, 8), (33, 64, 8, 4), (1, 8, 2, 2), (7, 128, 8, 8), (64, 16, 4, 2)])
def test_ep_plan_kernel_equals_restatement(ops, T, E, k, ep):
    """glm_ep_plan (one workgroup, device-side plan of the fixed-capacity dispatch) == its Python restatement
    (tests/ep_cpu_ops.ep_plan), every output array, for every rank; a capacity below the worst case reports overflow."""
    import ep_cpu_ops
    g = torch.Generator().manual_seed(T * 131 + E)
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]).to(torch.int64)
    w = (torch.rand(T, k, generator=g) + 0.1).to(BF)
    El, per = E // ep, (T + ep - 1) // ep
    cap = max(1, per * min(k, El))
    for r in range(ep):
        got = [t.cpu() for t in ops.ep_plan(dev(idx), dev(w), E, ep, r, cap)]
        exp = ep_cpu_ops.ep_plan(idx, w, E, ep, r, cap)
        lo, hi = min(T, r * per), min(T, (r + 1) * per)
        used = exp[0] >= 0
        assert torch.equal(got[0], exp[0]), f"rank {r}: send_gather"
        assert torch.equal(got[1][used], exp[1][used]), f"rank {r}: slot_scale"
        assert torch.equal(got[2][: (hi - lo) * k], exp[2][: (hi - lo) * k]), f"rank {r}: pair_slot"
        assert torch.equal(got[3], exp[3]) and torch.equal(got[4], exp[4]), f"rank {r}: receive side"
        assert int(got[5]) == 0
    if cap > 1:
        small = [t.cpu() for t in ops.ep_plan(dev(idx), dev(w), E, ep, 0, cap // 2 if cap // 2 >= 1 else 1)]
        exp = ep_cpu_ops.ep_plan(idx, w, E, ep, 0, cap // 2 if cap // 2 >= 1 else 1)
        assert int(small[5]) == int(exp[5]) and torch.equal(small[0], exp[0]) and torch.equal(small[4], exp[4])


def test_expert_parallel_fixed_dispatch_captures_in_a_graph(ops):
    """The fixed-capacity dispatch has no host read: ep = 4 logical ranks on one device, collectives by copies, the whole
    MoE block of all ranks captured in ONE HIP graph and replayed with new inputs == the eager run, bit for bit."""
    from nanovllm.distributed import expert_parallel as EP
    T, E, k, H, I, ep = 64, 32, 4, 512, 256, 4
    w13, w2 = dev(rnd(E + 1, 2 * I, H, seed=402, scale=H ** -0.5)), dev(rnd(E + 1, H, I, seed=403, scale=I ** -0.5))
    El = E // ep

    def shared(hh):
        a = ops.grouped_gemm_gateup_silu(hh, w13[E:E + 1], None, None, hh.shape[0])
        return ops.grouped_gemm_down(a, w2[E:E + 1], None, None, 0)
    xs, idxs, ws = dev(rnd(T, H, seed=404)), torch.zeros(T, k, dtype=torch.int64, device="cuda"), torch.zeros(T, k, dtype=BF, device="cuda")

    def block():
        gens = [EP.EPExperts(ops, ep, r, E, k, w13[r * El:(r + 1) * El], w2[r * El:(r + 1) * El], shared)
                .forward_fixed(xs, idxs, ws) for r in range(ep)]
        return EP.run_emulated(gens)[0]

    def new_inputs(seed):
        g = torch.Generator().manual_seed(seed)
        xs.copy_(rnd(T, H, seed=seed))
        idxs.copy_(torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)]))
        ws.copy_((torch.rand(T, k, generator=g) + 0.1).to(BF))
    new_inputs(1)
    block()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out_g = block()
    for seed in (2, 3, 4):
        new_inputs(seed)
        gr.replay()
        torch.cuda.synchronize()
        got = out_g.clone()
        assert torch.equal(got, block()), f"replay with inputs {seed} differs from the eager run"


def test_in_launch_handoffs_under_repetition(ops):
    """The router's two hand-offs (per-token logits gather, pair bucketing by the last workgroup) and the attention's
    kv-split merge read data that OTHER workgroups wrote in the same launch, from buffers that are reused launch after
    launch.  New inputs every repetition (a stale line from the previous launch would show), other kernels in between,
    results against the stand-alone kernels / the oracle every time."""
    T, H, E, k = 32, 4096, 64, 8
    nw = rnd(H, seed=702) * 0.1 + 1.0
    gw = rnd(E, H, seed=703, scale=H ** -0.5)
    bias = (torch.rand(E, generator=torch.Generator().manual_seed(704)) * 0.2 - 0.1).float()
    for rep in range(16):
        x, res = dev(rnd(T, H, seed=800 + rep)), dev(rnd(T, H, seed=900 + rep))
        y0, r0 = ops.add_rmsnorm(x, res, dev(nw), 1e-5)
        i0, w0 = ops.router_topk(y0, dev(gw), dev(bias), k, 1.0)
        y1, r1, i1, w1, al = ops.add_rmsnorm_router_topk(x, res, dev(nw), 1e-5, dev(gw), dev(bias), k, 1.0,
                                                         align_with_shared=True)
        ops.gemm_bf16(dev(rnd(32, 4096, seed=rep)), dev(rnd(2048, 4096, seed=1, scale=0.02)))  # keep other CUs busy
        assert torch.equal(i0, i1) and torch.equal(w0, w1) and torch.equal(y0, y1), f"rep {rep}"
        _check_align(*al, i1.cpu(), w1.cpu(), E, True)
    scale = 128 ** -0.5
    for rep in range(8):
        q, kc, vc, cl, bt = _paged_case([1024 - 37 * rep, 640 + rep], 96, 8, seed=1000 + rep)
        ref = O.attn_decode_paged(q, kc, vc, cl, bt, scale)
        out = ops.paged_attn_decode(dev(q), dev(kc), dev(vc), dev(cl), dev(bt), scale)  # 8 splits, in-kernel merge
        assert_mixed(out, ref, 1.6e-2, 4e-3, f"split merge rep {rep}")


# ------------------------------------------------------------------------------------------
# N3: one-shot all-reduce (opt-in; RCCL is the default path)
def test_oneshot_allreduce_world1_and_two_streams(ops):
    """Push-form one-shot all-reduce with the epoch in device memory.  world = 1; then two ranks emulated on ONE device
    (each "rank" is a stream, the peer pointers are the other rank's buffers): 7 calls (odd: both slots are reused out
    of phase), alternately one rank delayed by a long-running kernel in front of its call, so the early rank has to wait
    for the late one's flag — and the late one finds a flag that is already there."""
    from nanovllm.distributed.oneshot_allreduce import OneShotAllReduce
    dev0 = torch.device("cuda", 0)
    n = 32 * 4096
    ar = OneShotAllReduce(n, dev0, 0, 1)
    x = dev(rnd(32, 4096, seed=1101))
    ref = x.clone()
    for _ in range(3):
        y = x.clone()
        ar.all_reduce(y)
        assert torch.equal(y, ref)
    assert ar.status() == 0 and ar.epoch() == 3
    a0, a1 = OneShotAllReduce(n, dev0, 0, 2), OneShotAllReduce(n, dev0, 1, 2)
    for a in (a0, a1):
        a.peer_bufs, a.peer_flags = [a0.buf, a1.buf], [a0.flags, a1.flags]
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.randn(4096, 4096, device=dev0)
    for rep in range(7):
        x0, x1 = dev(rnd(32, 4096, seed=1200 + rep)), dev(rnd(32, 4096, seed=1300 + rep))
        exp = (x0.float() + x1.float()).to(BF)
        torch.cuda.synchronize()
        with torch.cuda.stream(s0):
            if rep % 3 == 1:
                (big @ big).sum()   # rank 0 arrives late
            a0.all_reduce(x0)
        with torch.cuda.stream(s1):
            if rep % 3 == 2:
                (big @ big).sum()   # rank 1 arrives late
            a1.all_reduce(x1)
        torch.cuda.synchronize()
        assert a0.status() == 0 and a1.status() == 0, "a rank gave up waiting (the two kernels did not overlap?)"
        assert torch.equal(x0, exp) and torch.equal(x1, exp), f"rep {rep}"
    assert a0.epoch() == 7 and a1.epoch() == 7
    # smaller tensors reuse the same slots
    for rep in range(3):
        x0, x1 = dev(rnd(2, 4096, seed=1400 + rep)), dev(rnd(2, 4096, seed=1500 + rep))
        exp = (x0.float() + x1.float()).to(BF)
        with torch.cuda.stream(s0):
            a0.all_reduce(x0)
        with torch.cuda.stream(s1):
            a1.all_reduce(x1)
        torch.cuda.synchronize()
        assert torch.equal(x0, exp) and torch.equal(x1, exp)


def test_comm_blocks_are_parked_not_freed(ops):
    """glm_comm_free parks a block for the next glm_comm_alloc of the same device, kind and size instead of returning it to
    the runtime (r04: hipFree'd MiB-sized uncached blocks made ordinary tensors allocated afterwards read stale,
    profiles/r04/c1_stale_readback.txt): the same pointer comes back, zero-filled; a pointer the library did not hand out is
    refused."""
    import ctypes
    from nanovllm.distributed.oneshot_allreduce import CommBuffer
    lib = ops.load_library()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    d2d = 3  # hipMemcpyDeviceToDevice
    a = CommBuffer(3 << 20, "cuda:0")
    pa = a.data_ptr()
    ones = torch.arange(1, 17, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert hip.hipMemcpy(pa, ones.data_ptr(), 64, d2d) == 0        # dirty the block
    a.close()
    b_ = CommBuffer(3 << 20, "cuda:0")
    assert b_.data_ptr() == pa, "a parked block of the same size must be reused"
    probe = torch.full((16,), 7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    assert hip.hipMemcpy(probe.data_ptr(), b_.data_ptr(), 64, d2d) == 0
    torch.cuda.synchronize()
    assert not bool(probe.any()), "a reused block is zero-filled like a fresh one"
    c_ = CommBuffer(1 << 20, "cuda:0")
    assert c_.data_ptr() != pa
    b_.close()
    c_.close()
    rc = lib.glm_comm_free(ctypes.c_void_p(probe.data_ptr()))
    assert rc == -1 and b"not allocated by glm_comm_alloc" in lib.glm_last_error()  # GLM_ERR_INVALID


def test_oneshot_allreduce_missing_peer_is_loud(ops):
    """A peer that never arrives: the bounded wait expires, the output is NaN (never stale data), status names the rank,
    check() raises — and the epoch still advances so a later call pairs up again."""
    from nanovllm.distributed.oneshot_allreduce import OneShotAllReduce
    dev0 = torch.device("cuda", 0)
    n = 8 * 1024
    a0, a1 = OneShotAllReduce(n, dev0, 0, 2), OneShotAllReduce(n, dev0, 1, 2)
    for a in (a0, a1):
        a.peer_bufs, a.peer_flags = [a0.buf, a1.buf], [a0.flags, a1.flags]
    x0 = dev(rnd(8, 1024, seed=1600))
    a0.all_reduce(x0)          # rank 1 never calls
    torch.cuda.synchronize()
    assert a0.status() == 2 and bool(torch.isnan(x0.float()).all())
    with pytest.raises(RuntimeError, match="gave up waiting for rank 1"):
        a0.check()
    assert a0.epoch() == 1