"""Parity tests of the STUDY kernels (measured against the shipped ones and lost): prefill attention
on 32x32x16 MFMAs (option fa_mfma32, r03), with 64 query rows per wave (option fa_rows64, csrc/prefill_attn64.hip, r04) and as
persistent workgroups with a run-time block queue (option fa_persist, csrc/prefill_attn_persist.hip, r05).

Not collected by `pytest tests/` (the file name does not match): these kernels are not in the product library.  Run by
tests/test_study_kernels_gpu.py in a child process against lib/libglm4moe_hip_study.so when that library has been built
(`python glm4_moe_nano-vllm_amd/build.py --study`), or by hand:

    GLM4MOE_HIP_LIB=$PWD/glm4_moe_nano-vllm_amd/lib/libglm4moe_hip_study.so python -m pytest tests/study_kernels.py -q
"""
import pytest
import torch

from oracle import glm4_moe_oracle as O
from test_hip_kernels import BF, _paged_case, assert_mixed, dev, ops, opt_guard  # noqa: F401 - fixtures

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lens_q,lens_k,nh,nkv,paged", [
    ([300, 17, 513, 64], [300, 17, 513, 64], 16, 4, False), ([100, 50], [612, 306], 16, 4, True),
    ([1025], [1025], 12, 1, False), ([5, 40], [300, 296], 4, 2, True), ([33], [33], 8, 2, False)])
def test_varlen_attn_prefill_mfma32(ops, opt_guard, lens_q, lens_k, nh, nkv, paged):
    """The opt-in 32x32x16 kernel (GLM_FA_MFMA32=1, read at every call) against the oracle on the varlen cases, and against
    the default kernel's output (two bf16 evaluations of the same attention: same tolerance)."""
    g = torch.Generator().manual_seed(sum(lens_k) + nh)
    hd, bs = 128, 256
    q = torch.randn(sum(lens_q), nh, hd, generator=g).to(BF)
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32)
    scale = hd ** -0.5
    if paged:
        _, kc, vc, _, bt = _paged_case(lens_k, nh, nkv, seed=5)
        ref = O.attn_prefill_varlen(q, kc, vc, cu_q, cu_k, scale, bt)
        run = lambda: ops.varlen_attn_prefill(dev(q), dev(kc), dev(vc), dev(cu_q), dev(cu_k), max(lens_q), scale, dev(bt), bs)
    else:
        k = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF)
        v = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF)
        ref = O.attn_prefill_varlen(q, k, v, cu_q, cu_k, scale)
        run = lambda: ops.varlen_attn_prefill(dev(q), dev(k), dev(v), dev(cu_q), dev(cu_k), max(lens_q), scale)
    base = run()
    opt_guard("fa_mfma32", 1)
    out = run()
    assert torch.isfinite(out.float()).all()
    assert_mixed(out, ref, 1.6e-2, 4e-3, f"prefill attn (32x32x16) q={lens_q} k={lens_k}")
    assert_mixed(out, base, 1.6e-2, 4e-3, "32x32x16 kernel vs default kernel")


@pytest.mark.parametrize("lens_q,lens_k,nh,nkv,paged", [
    ([300, 17, 513, 64], [300, 17, 513, 64], 16, 4, False), ([100, 50], [612, 306], 16, 4, True),
    ([1025], [1025], 12, 1, False), ([5, 40], [300, 296], 4, 2, True), ([33], [33], 8, 2, False),
    ([2048, 700], [2048, 700], 8, 2, False), ([600], [1900], 4, 1, True), ([1], [1], 2, 1, False)])
def test_varlen_attn_prefill_rows64(ops, opt_guard, lens_q, lens_k, nh, nkv, paged):
    """The opt-in kernel with 64 query rows per wave (option fa_rows64, csrc/prefill_attn64.hip) against the oracle
    (attention.py:59-75 = flash_attn_varlen_func, causal) on the varlen cases — ragged, chunked prefill over a paged cache,
    and lengths whose key loop reaches the predicate-free interior tiles — and against the default kernel's output."""
    g = torch.Generator().manual_seed(sum(lens_k) + nh)
    hd, bs = 128, 256
    q = torch.randn(sum(lens_q), nh, hd, generator=g).to(BF)
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32)
    scale = hd ** -0.5
    if paged:
        _, kc, vc, _, bt = _paged_case(lens_k, nh, nkv, seed=5)
        ref = O.attn_prefill_varlen(q, kc, vc, cu_q, cu_k, scale, bt)
        run = lambda: ops.varlen_attn_prefill(dev(q), dev(kc), dev(vc), dev(cu_q), dev(cu_k), max(lens_q), scale, dev(bt), bs)
    else:
        k = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF)
        v = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF)
        ref = O.attn_prefill_varlen(q, k, v, cu_q, cu_k, scale)
        run = lambda: ops.varlen_attn_prefill(dev(q), dev(k), dev(v), dev(cu_q), dev(cu_k), max(lens_q), scale)
    base = run()
    opt_guard("fa_rows64", 1)
    out = run()
    assert torch.isfinite(out.float()).all()
    assert_mixed(out, ref, 1.6e-2, 4e-3, f"prefill attn (64 rows per wave) q={lens_q} k={lens_k}")
    assert_mixed(out, base, 1.6e-2, 4e-3, "64-row kernel vs default kernel")


@pytest.mark.parametrize("lens_q,lens_k,nh,nkv", [
    ([21, 16], [21, 16], 4, 2), ([64], [64], 2, 1), ([65, 1, 200], [65, 1, 200], 12, 2),
    ([300, 17, 513, 64], [300, 17, 513, 64], 16, 4),            # XCD-grouped order, several blocks per sequence
    ([5, 40], [300, 296], 4, 2), ([130], [513], 8, 8),          # fewer queries than keys (contiguous prefix)
    ([200, 70], [40, 70], 4, 2),                                # more queries than keys: rows that see nothing
    ([1024] * 5, [1024] * 5, 16, 2),                            # 640 blocks on 512 slots: two rounds, plain order
    ([2048, 100, 1, 700] * 2, [2048, 100, 1, 700] * 2, 8, 4),   # 1024 blocks, XCD order, ragged: empty blocks skipped
    ([4096, 3000], [4096, 3000], 32, 8)])                       # four rounds per slot (bit-identity only)
def test_varlen_attn_prefill_persist_bit_identical(ops, opt_guard, lens_q, lens_k, nh, nkv):
    """r05: the persistent-workgroup form of the contiguous-K/V prefill attention (option fa_persist,
    csrc/prefill_attn_persist.hip) — blocks walked by resident workgroups, K/V tiles of consecutive blocks as one stream
    through the LDS ring — does the same arithmetic per (query row, key) as the one-block-per-workgroup kernel: outputs
    are bit-identical; the smaller cases are checked against the oracle as well (attention.py:97-100)."""
    g = torch.Generator().manual_seed(sum(lens_k) + nh)
    hd = 128
    q = torch.randn(sum(lens_q), nh, hd, generator=g).to(BF)
    k = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF)
    v = torch.randn(sum(lens_k), nkv, hd, generator=g).to(BF)
    cu_q = torch.tensor([0] + list(torch.tensor(lens_q).cumsum(0)), dtype=torch.int32)
    cu_k = torch.tensor([0] + list(torch.tensor(lens_k).cumsum(0)), dtype=torch.int32)
    scale = hd ** -0.5
    args = (dev(q), dev(k), dev(v), dev(cu_q), dev(cu_k), max(lens_q), scale)
    opt_guard("fa_persist", 0)
    base = ops.varlen_attn_prefill(*args)
    opt_guard("fa_persist", 1)
    outs = [ops.varlen_attn_prefill(*args) for _ in range(2)]
    torch.cuda.synchronize()
    for o in outs:
        assert torch.isfinite(o.float()).all()
        assert torch.equal(o, base), f"persistent prefill attention differs from the default kernel: q={lens_q} k={lens_k}"
    if sum(a * b for a, b in zip(lens_q, lens_k)) * nh <= 2.5e8:
        ref = O.attn_prefill_varlen(q, k, v, cu_q, cu_k, scale)
        assert_mixed(outs[0], ref, 1.6e-2, 4e-3, f"persistent prefill attn q={lens_q} k={lens_k}")
