"""CPU checks of the W8A8 oracle leg (oracle/glm4_moe_oracle.py, r05): the two quantisers restated there and the record layout
of the E8M0 block scales.  Parity of this leg is unpinned by construction (the reference has no fp8 path); what CAN be pinned
on the CPU is that the restatement is self-consistent: quantise -> dequantise stays within half an e4m3 step, nothing saturates,
the scale is the smallest power of two that keeps the block inside +-448, and the C-ABI's [R, 64] scale record round-trips."""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import glm4_moe_oracle as O  # noqa: E402


def test_mx_quantiser_bounds_and_scale_choice():
    g = torch.Generator().manual_seed(3)
    v = (torch.randn(257, 1408, generator=g) * torch.logspace(-6, 3, 257).unsqueeze(-1)).to(torch.bfloat16)
    v[0] = 0
    v[1, :32] = 0
    q, sb = O.quantize_mxfp8(v)
    assert q.dtype == torch.float8_e4m3fn and sb.dtype == torch.uint8 and sb.shape == (257, 44)
    qf = q.to(torch.float32).view(257, 44, 32)
    assert float(qf.abs().max()) <= 448.0 and not bool(torch.isnan(qf).any())
    amax = v.float().view(257, 44, 32).abs().amax(-1)
    scale = torch.exp2(sb.float() - 127.0)
    nz = amax > 0
    # the scaled block maximum lies in (224, 448]: the smallest power-of-two scale that does not saturate
    sm = (amax / scale)[nz]
    assert bool((sm <= 448.0).all()) and bool((sm > 224.0).all()), (float(sm.min()), float(sm.max()))
    assert bool((sb[~nz] == 127).all())
    d = O.dequant_mxfp8(q, sb)
    err = (d - v.float()).abs().view(257, 44, 32)
    # RNE: half a step of the element's binade (2^-4 relative) for normals, 2^-10 of the scaled unit for e4m3 subnormals
    lim = 2 ** -4 * v.float().abs().view(257, 44, 32) + (2 ** -10 * scale).unsqueeze(-1)
    assert bool((err <= lim * (1 + 1e-6)).all())


def test_scale_record_layout_roundtrip():
    sb = torch.arange(3 * 44, dtype=torch.int32).view(3, 44).to(torch.uint8)
    rec = O.mx_scale_record(sb)
    assert rec.shape == (3, 64)
    # block b = 4 kt + g sits at byte g * 16 + kt: the down kernel's lane row g reads the bytes of K tiles 0..15 contiguously
    for b in (0, 1, 5, 43):
        assert int(rec[1, (b & 3) * 16 + (b >> 2)]) == int(sb[1, b])
    assert torch.equal(O.mx_scale_unrecord(rec, 44), sb)


def test_row_quantiser_and_gemm_legs():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(7, 256, generator=g).to(torch.bfloat16)
    x[3] = 0
    xq, xs = O.quantize_fp8_rows(x)
    assert float(xq.to(torch.float32).abs().max()) <= 448.0
    back = xq.to(torch.float32) * xs.unsqueeze(-1)
    assert bool(((back - x.float()).abs() <= 2 ** -4 * x.float().abs() + 2 ** -9 * xs.unsqueeze(-1)).all())
    w13 = (torch.randn(2 * 64, 256, generator=g) * 0.05).to(torch.bfloat16)
    w2 = (torch.randn(256, 64, generator=g) * 0.05).to(torch.bfloat16)
    q13, s13 = O.quantize_fp8_rowwise(w13)
    q2, s2 = O.quantize_fp8_rowwise(w2)
    act = O.gateup_silu_fp8a8(xq, xs, q13, s13)
    assert act.shape == (7, 64) and act.dtype == torch.bfloat16
    aq, asb = O.quantize_mxfp8(act)
    y = O.down_fp8a8(aq, asb, q2, s2)
    # against the weight-only leg (bf16 activations): the activation quantisation costs a few per cent, not more
    y_w = O.linear_fp8w(O.silu_and_mul(O.linear_fp8w(x, q13, s13)), q2, s2)
    rel = float((y.float() - y_w.float()).norm() / y_w.float().norm().clamp_min(1e-30))
    assert rel < 0.08, rel
