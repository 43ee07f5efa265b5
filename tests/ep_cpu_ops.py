"""Test-side stand-in for nanovllm.hip_ops with the ORACLE's arithmetic on CPU tensors, so that the expert-parallel
index / permutation / collective logic (distributed/expert_parallel.py) can run over gloo without a GPU.  Only the
tests import this; the product has no CPU path."""
import torch

from oracle import glm4_moe_oracle as O


def moe_align(topk_idx, topk_w, E, with_shared):
    assert not with_shared
    T, k = topk_idx.shape
    flat = topk_idx.reshape(-1)
    order = torch.sort(flat, stable=True)[1]                      # pairs grouped by expert, ascending expert id
    counts = torch.bincount(flat, minlength=E)
    offs = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)]).to(torch.int32)
    tok = (order // k).to(torch.int32)
    scale = topk_w.reshape(-1)[order].contiguous()
    pos = torch.empty(T * k, dtype=torch.int32)
    pos[order] = torch.arange(T * k, dtype=torch.int32)
    return offs, tok, scale, pos


def gather_rows(src, idx):
    out = torch.full((idx.numel(), src.shape[1]), float("nan"), dtype=src.dtype)  # unused rows: poison, nobody may read them
    ok = idx >= 0
    out[ok] = src[idx[ok].long()]
    return out


def scatter_rows(src, idx, n_rows=None, no_init=False):
    fill = float("nan") if no_init else 0.0
    dst = torch.full((n_rows if n_rows is not None else src.shape[0], src.shape[1]), fill, dtype=src.dtype)
    ok = idx >= 0
    dst[idx[ok].long()] = src[ok]
    return dst


def ep_plan(topk_idx, topk_w, E, ep, rank, cap):
    """Python restatement of glm_ep_plan (csrc/moe_misc.hip): same slot layout, same order inside a slot."""
    T, k = topk_idx.shape
    El, per = E // ep, (T + ep - 1) // ep
    lo, hi = min(T, rank * per), min(T, (rank + 1) * per)
    Tl = hi - lo
    cnt = torch.zeros((ep, E), dtype=torch.long)
    for t in range(T):
        for j in range(k):
            cnt[t // per, int(topk_idx[t, j])] += 1
    send_gather = torch.full((ep * cap,), -1, dtype=torch.int32)
    slot_scale = torch.zeros(ep * cap, dtype=topk_w.dtype)
    pair_slot = torch.full((max(Tl * k, 1),), -1, dtype=torch.int32)
    fill = [0] * ep
    for e in range(E):                       # expert ascending, then pair index ascending
        d = e // El
        for p in range(Tl * k):
            t, j = lo + p // k, p % k
            if int(topk_idx[t, j]) != e:
                continue
            if fill[d] < cap:
                slot = d * cap + fill[d]
                send_gather[slot], slot_scale[slot], pair_slot[p] = t, topk_w[t, j], slot
            fill[d] += 1
    recv_offs = torch.zeros(El + 1, dtype=torch.int32)
    recv_gidx = torch.full((ep * cap,), -1, dtype=torch.int32)
    row = 0
    for el in range(El):
        recv_offs[el] = row
        for s in range(ep):
            within = int(cnt[s, rank * El:rank * El + el].sum())
            n = max(0, min(int(cnt[s, rank * El + el]), cap - within))
            for i in range(n):
                recv_gidx[row] = s * cap + within + i
                row += 1
    recv_offs[El] = row
    worst = int(cnt.view(ep, ep, El).sum(-1).max())
    return send_gather, slot_scale, pair_slot, recv_offs, recv_gidx, torch.tensor([worst if worst > cap else 0], dtype=torch.int32)


def grouped_gemm_gateup_silu(x, w13, offs, src_idx, R):
    I = w13.shape[1] // 2
    act = torch.zeros((R, I), dtype=x.dtype)
    for g in range(w13.shape[0]):
        lo, hi = int(offs[g]), int(offs[g + 1])
        if hi > lo:
            rows = x[src_idx[lo:hi].long()] if src_idx is not None else x[lo:hi]
            act[lo:hi] = O.silu_and_mul(O.linear(rows, w13[g]))
    return act


def grouped_gemm_down(act, w2, offs, scale, n_scaled):
    assert n_scaled == 0
    y = torch.zeros((act.shape[0], w2.shape[1]), dtype=act.dtype)
    for g in range(w2.shape[0]):
        lo, hi = int(offs[g]), int(offs[g + 1])
        if hi > lo:
            y[lo:hi] = O.linear(act[lo:hi], w2[g])
    return y


def grouped_gemm_gateup_silu_fp8w(x, w13_q, w13_s, offs, src_idx, R):
    """fp8 e4m3 weight-only experts (oracle arithmetic: fp32 accumulation, row scale on the sum, one rounding)."""
    I = w13_q.shape[1] // 2
    act = torch.zeros((R, I), dtype=x.dtype)
    if offs is None:  # one group over all rows (the shared expert)
        return O.silu_and_mul(O.linear_fp8w(x, w13_q[0], w13_s[0]))
    for g in range(w13_q.shape[0]):
        lo, hi = int(offs[g]), int(offs[g + 1])
        if hi > lo:
            rows = x[src_idx[lo:hi].long()] if src_idx is not None else x[lo:hi]
            act[lo:hi] = O.silu_and_mul(O.linear_fp8w(rows, w13_q[g], w13_s[g]))
    return act


def grouped_gemm_down_fp8w(act, w2_q, w2_s, offs, scale, n_scaled):
    assert n_scaled == 0
    if offs is None:
        return O.linear_fp8w(act, w2_q[0], w2_s[0])
    y = torch.zeros((act.shape[0], w2_q.shape[1]), dtype=act.dtype)
    for g in range(w2_q.shape[0]):
        lo, hi = int(offs[g]), int(offs[g + 1])
        if hi > lo:
            y[lo:hi] = O.linear_fp8w(act[lo:hi], w2_q[g], w2_s[g])
    return y


def grouped_gemm_gateup_silu_cap(x, w13, offs, src_idx, R_cap, R_expected):
    return grouped_gemm_gateup_silu(x, w13, offs, src_idx, R_cap)


def grouped_gemm_down_cap(act, w2, offs, R_expected):
    return grouped_gemm_down(act, w2, offs, None, 0)


def moe_combine_scaled(y, pair_pos, row_scale, shared_rows, T, k):
    out = torch.zeros((T, y.shape[1]), dtype=y.dtype)
    pp = pair_pos.view(T, k)
    for t in range(T):
        acc = torch.zeros(y.shape[1], dtype=y.dtype)
        for p in sorted(int(v) for v in pp[t].tolist() if v >= 0):   # ascending row == ascending expert id
            acc = acc + y[p] * row_scale[p]                           # bf16 multiply, bf16 add (moe.py:149-151)
        out[t] = acc + shared_rows[t] if shared_rows is not None else acc
    return out
