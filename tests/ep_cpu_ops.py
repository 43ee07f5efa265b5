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
    return src[idx.long()].contiguous()


def scatter_rows(src, idx, n_rows=None):
    dst = torch.zeros((n_rows if n_rows is not None else src.shape[0], src.shape[1]), dtype=src.dtype)
    dst[idx.long()] = src
    return dst


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


def moe_combine_scaled(y, pair_pos, row_scale, shared_rows, T, k):
    out = torch.zeros((T, y.shape[1]), dtype=y.dtype)
    pp = pair_pos.view(T, k)
    for t in range(T):
        acc = torch.zeros(y.shape[1], dtype=y.dtype)
        for p in sorted(int(v) for v in pp[t].tolist() if v >= 0):   # ascending row == ascending expert id
            acc = acc + y[p] * row_scale[p]                           # bf16 multiply, bf16 add (moe.py:149-151)
        out[t] = acc + shared_rows[t] if shared_rows is not None else acc
    return out
