"""Parity tests of the STUDY kernels (measured against the shipped ones and lost): prefill attention
on 32x32x16 MFMAs (option fa_mfma32, r03) and with 64 query rows per wave (option fa_rows64, csrc/prefill_attn64.hip, r04).

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
