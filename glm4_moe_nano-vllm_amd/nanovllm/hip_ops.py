"""ctypes binding of libglm4moe_hip.so (include/glm4moe_hip.h) for torch tensors on ROCm.

This is the only place where Python touches the C-ABI: raw ``data_ptr()`` values plus the
current torch stream go in, nothing comes back but a status code.  There is NO CPU or eager
fallback: if the shared object is missing, or a tensor is not on the GPU, the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(
    "GLM4MOE_HIP_LIB", os.path.join(_HERE, "..", "lib", "libglm4moe_hip.so"))

_lib = None

# name -> (restype, argtypes); mirrors include/glm4moe_hip.h one to one
_SIGNATURES = {
    "glm_abi_version": (c_int, []),
    "glm_last_error": (c_char_p, []),
    "glm_get_option": (c_int, [c_char_p, c_void_p]),
    "glm_set_option": (c_int, [c_char_p, c_int]),
    "glm_option_name": (c_char_p, [c_int]),
    "glm_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_void_p]),
    "glm_add_rmsnorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                c_float, c_void_p]),
    "glm_head_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int,
                                 c_int, c_float, c_void_p]),
    "glm_rope_neox": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                              c_int64, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                              c_int, c_void_p]),
    "glm_kv_store": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                             c_int64, c_int, c_int64, c_void_p]),
    "glm_qkv_split_norm_rope_store": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                              c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                              c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "glm_gemm_qkv_rope_store": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                        c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_void_p]),
    "glm_silu_and_mul": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "glm_paged_attn_decode_workspace_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "glm_paged_attn_decode_splits": (c_int, [c_int64, c_int, c_int, c_int64, c_int, c_int]),
    "glm_paged_attn_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                      c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int64,
                                      c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "glm_paged_attn_decode_qkv": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                          c_int64, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "glm_paged_attn_decode_qkv_partial": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p,
                                                  c_int64, c_int, c_int, c_int, c_int, c_int64, c_float, c_int, c_int,
                                                  c_void_p, c_size_t, c_void_p]),
    "glm_varlen_attn_prefill": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int,
                                        c_int, c_int, c_float, c_void_p]),
    "glm_router_topk": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_int64, c_int, c_int, c_int, c_float, c_void_p]),
    "glm_moe_align_workspace_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "glm_moe_align_parallel": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                       c_int, c_void_p, c_size_t, c_void_p]),
    "glm_router_select": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p]),
    "glm_moe_align": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                              c_int, c_int, c_int, c_void_p]),
    "glm_grouped_gemm_gateup_silu": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                             c_void_p, c_void_p, c_int64, c_int, c_int, c_int,
                                             c_void_p]),
    "glm_grouped_gemm_down": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                      c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "glm_fp8w_expand_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "glm_grouped_gemm_gateup_silu_fp8w": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                                  c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int,
                                                  c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "glm_grouped_gemm_down_fp8w": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p,
                                           c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                           c_void_p, c_size_t, c_void_p]),
    "glm_moe_combine": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_int,
                                c_int64, c_void_p]),
    "glm_gemm_bf16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                              c_int64, c_int, c_int, c_void_p]),
    "glm_grouped_gemm_gateup_silu_fewtok": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                                    c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "glm_grouped_gemm_down_fewtok": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int,
                                             c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "glm_gemm_bf16_combine_norm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int64, c_void_p, c_void_p,
                                           c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int64,
                                           c_int, c_int, c_void_p]),
    "glm_gemm_bf16_router_tail": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "glm_quantize_fp8_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "glm_grouped_gemm_gateup_silu_fp8a8": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                                   c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                                                   c_int, c_void_p]),
    "glm_grouped_gemm_down_fp8a8": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                                            c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                            c_void_p]),
    "glm_decode_mlp_engine_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "glm_decode_mlp_engine": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                                      c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "glm_gemm_bf16_partial_slices": (c_int, [c_int, c_int]),
    "glm_gemm_bf16_partial": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                                      c_int, c_void_p]),
    "glm_add_rmsnorm_router_topk": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_float, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                            c_int, c_float, c_void_p]),
    "glm_add_rmsnorm_router_topk_align": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_float,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                                  c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                                  c_int, c_void_p, c_size_t, c_void_p]),
    "glm_router_workspace_bytes": (c_size_t, [c_int64, c_int]),
    "glm_allreduce_oneshot_buffer_elems": (c_size_t, [c_int, c_int64]),
    "glm_comm_alloc": (c_int, [c_void_p, c_size_t, c_int]),
    "glm_comm_free": (c_int, [c_void_p]),
    "glm_comm_ipc_export": (c_int, [c_void_p, c_void_p]),
    "glm_comm_ipc_open": (c_int, [c_void_p, c_void_p]),
    "glm_comm_ipc_close": (c_int, [c_void_p]),
    "glm_allreduce_oneshot": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_void_p,
                                      c_void_p]),
    "glm_ep_plan": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                            c_void_p, c_void_p, c_void_p, c_void_p]),
    "glm_grouped_gemm_gateup_silu_cap": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p,
                                                 c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "glm_grouped_gemm_down_cap": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int, c_void_p,
                                          c_int64, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "glm_gather_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "glm_scatter_rows": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "glm_moe_combine_scaled": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                       c_int, c_void_p]),
    "glm_moe_combine_add_rmsnorm": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_int, c_int64, c_void_p,
                                            c_void_p, c_void_p, c_void_p, c_float, c_void_p]),
    "glm_add_rmsnorm_partial": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int64, c_int, c_float, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load_library(path: str | None = None) -> ctypes.CDLL:
    """dlopen the HIP library and attach prototypes; raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = os.path.abspath(path or LIB_PATH)
    if not os.path.exists(p):
        raise RuntimeError(
            f"libglm4moe_hip.so not found at {p}: build it with "
            "`python glm4_moe_nano-vllm_amd/build.py` (there is no CPU fallback)")
    lib = ctypes.CDLL(p)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
        _read_host_options()
    return lib


# ---- options -----------------------------------------------------------------------------------------------------------
# Host-side switches of the launch paths (which fused launch a layer takes, which collective form): read from the
# environment ONCE, when the library is loaded, like the library's own table (glm_get_option / glm_set_option).  A path never
# consults os.environ afterwards; tests and A/B tools call set_option().  options() is the effective set of both sides
# (bench.py prints it in its line's config).
_HOST_OPTION_ENV = {  # name -> (environment variable, default)
    "fuse_combine_qkv": ("GLM4MOE_FUSE_COMBINE_QKV", 1),    # one token: combine + input norm + QKV in one launch
    "fuse_oproj_router": ("GLM4MOE_FUSE_OPROJ_ROUTER", 1),  # one token: o_proj + norm + router + bucketing in one launch
    "defer_qkv_reduce": ("GLM4MOE_DEFER_QKV_REDUCE", 1),    # QKV as K-slice partial sums reduced by the attention's loads
    "fuse_qkv_rope": ("GLM4MOE_FUSE_QKV_ROPE", 1),          # prefill: split + rotary + K/V store in the QKV GEMM's epilogue
    "t1_engine": ("GLM4MOE_T1_ENGINE", 0),                  # one token: o_proj .. down projection as ONE persistent launch (r05; opt-in:
                                                            # bit-identical, 5-6 us per layer slower — profiles/r05/decode_engine_stamps.txt)
    "ep_fixed": ("GLM_EP_FIXED", 1),                        # expert parallelism: fixed-capacity dispatch at decode sizes
    "oneshot_allreduce": ("GLM_ONESHOT_ALLREDUCE", 0),      # opt-in one-hop all-reduce over mapped peer buffers
}
_host_opts: dict = {}


def _read_host_options() -> None:
    for name, (env, dflt) in _HOST_OPTION_ENV.items():
        v = os.environ.get(env)
        _host_opts[name] = int(v) if v not in (None, "") else dflt


def opt(name: str) -> int:
    """Effective value of a launch-path option (host-side table or the library's)."""
    if not _host_opts:
        load_library()
    if name in _host_opts:
        return _host_opts[name]
    v = c_int(0)
    _check(load_library().glm_get_option(name.encode(), ctypes.byref(v)), "glm_get_option")
    return v.value


def set_option(name: str, value: int) -> int:
    """Change an option for the launches that follow (NOT for HIP graphs captured earlier); returns the old value."""
    old = opt(name)
    if name in _host_opts:
        _host_opts[name] = int(value)
    else:
        _check(load_library().glm_set_option(name.encode(), int(value)), "glm_set_option")
    return old


class option:
    """`with hip_ops.option("tinym", 0): ...`"""

    def __init__(self, name: str, value: int):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def options() -> dict:
    lib = load_library()
    out, i = {}, 0
    while True:
        n = lib.glm_option_name(i)
        if not n:
            break
        out[n.decode()] = opt(n.decode())
        i += 1
    out.update(_host_opts)
    return out


GLM_ERR_UNSUPPORTED = -2  # include/glm4moe_hip.h


def _check(rc: int, name: str) -> None:
    if rc != 0:
        msg = load_library().glm_last_error()
        raise RuntimeError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("glm4moe HIP ops need GPU tensors (no CPU fallback)")
    return t.data_ptr()


def _bf16(t: torch.Tensor, name: str) -> None:
    if t.dtype != torch.bfloat16:
        raise TypeError(f"{name}: expected bfloat16, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: innermost stride must be 1")


# ------------------------------------------------------------------------------------------
# norms
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    _bf16(x, "x"); _bf16(weight, "weight")
    H = x.shape[-1]
    x2 = x.reshape(-1, H)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    y = torch.empty_like(x2)
    _check(load_library().glm_rmsnorm(_p(x2), _p(weight), _p(y), x2.shape[0], H, eps, _stream()),
           "glm_rmsnorm")
    return y.view(x.shape)


def add_rmsnorm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    _bf16(x, "x"); _bf16(residual, "residual"); _bf16(weight, "weight")
    H = x.shape[-1]
    x2 = x.reshape(-1, H).contiguous()
    r2 = residual.reshape(-1, H).contiguous()
    y = torch.empty_like(x2)
    ro = torch.empty_like(x2)
    _check(load_library().glm_add_rmsnorm(_p(x2), _p(r2), _p(weight), _p(y), _p(ro), x2.shape[0],
                                          H, eps, _stream()), "glm_add_rmsnorm")
    return y.view(x.shape), ro.view(x.shape)


def head_rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """x: [T, n_heads, hd] (head stride hd, arbitrary token stride) -> contiguous result."""
    _bf16(x, "x"); _bf16(weight, "weight")
    T, nh, hd = x.shape
    if x.stride(1) != hd:
        x = x.contiguous()
    y = torch.empty((T, nh, hd), dtype=x.dtype, device=x.device)
    _check(load_library().glm_head_rmsnorm(_p(x), x.stride(0), _p(weight), _p(y), nh * hd, T, nh,
                                           hd, eps, _stream()), "glm_head_rmsnorm")
    return y


# ------------------------------------------------------------------------------------------
# rotary / kv cache
def rope_neox(positions: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
              cos_sin_cache: torch.Tensor, rotary_dim: int) -> tuple[torch.Tensor, torch.Tensor]:
    """q: [T,nh,hd], k: [T,nkv,hd] (head stride hd) -> new contiguous (q, k)."""
    _bf16(q, "q"); _bf16(k, "k")
    T, nh, hd = q.shape
    nkv = k.shape[1]
    if q.stride(1) != hd:
        q = q.contiguous()
    if k.stride(1) != hd:
        k = k.contiguous()
    assert positions.dtype == torch.int64 and cos_sin_cache.dtype == torch.float32
    cs = cos_sin_cache.reshape(cos_sin_cache.shape[0], -1)
    assert cs.is_contiguous() and cs.shape[1] == rotary_dim
    qo = torch.empty((T, nh, hd), dtype=q.dtype, device=q.device)
    ko = torch.empty((T, nkv, hd), dtype=k.dtype, device=k.device)
    _check(load_library().glm_rope_neox(_p(q), q.stride(0), _p(k), k.stride(0), _p(qo), nh * hd,
                                        _p(ko), nkv * hd, _p(positions.contiguous()), _p(cs),
                                        cs.shape[0], T, nh, nkv, hd, rotary_dim, _stream()),
           "glm_rope_neox")
    return qo, ko


def kv_store(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
             slot_mapping: torch.Tensor) -> None:
    _bf16(key, "key"); _bf16(value, "value")
    N, nh, hd = key.shape
    D = nh * hd
    assert key.stride(1) == hd and value.stride(1) == hd
    assert k_cache.stride(1) == D and v_cache.stride(1) == D
    assert slot_mapping.numel() == N and slot_mapping.dtype == torch.int32
    n_slots = k_cache.numel() // D
    _check(load_library().glm_kv_store(_p(key), key.stride(0), _p(value), value.stride(0),
                                       _p(k_cache), _p(v_cache), _p(slot_mapping), N, D, n_slots,
                                       _stream()), "glm_kv_store")


def qkv_split_norm_rope_store(qkv: torch.Tensor, nh: int, nkv: int, hd: int,
                              positions: torch.Tensor, cos_sin_cache: torch.Tensor,
                              rotary_dim: int, q_norm_w: torch.Tensor | None,
                              k_norm_w: torch.Tensor | None, eps: float,
                              k_cache: torch.Tensor | None, v_cache: torch.Tensor | None,
                              slot_mapping: torch.Tensor | None, need_kv_out: bool):
    """Fused split + (qk-norm) + rotary + cache scatter; returns (q, k|None, v|None)."""
    _bf16(qkv, "qkv")
    T = qkv.shape[0]
    assert qkv.shape[1] == (nh + 2 * nkv) * hd
    cs = cos_sin_cache.reshape(cos_sin_cache.shape[0], -1)
    q = torch.empty((T, nh, hd), dtype=qkv.dtype, device=qkv.device)
    k = torch.empty((T, nkv, hd), dtype=qkv.dtype, device=qkv.device) if need_kv_out else None
    v = torch.empty((T, nkv, hd), dtype=qkv.dtype, device=qkv.device) if need_kv_out else None
    store = k_cache is not None and k_cache.numel() > 0 and slot_mapping is not None
    n_slots = (k_cache.numel() // (nkv * hd)) if store else 0
    _check(load_library().glm_qkv_split_norm_rope_store(
        _p(qkv), qkv.stride(0), _p(q), _p(k), _p(v), _p(q_norm_w), _p(k_norm_w), eps,
        _p(positions), _p(cs), cs.shape[0], rotary_dim,
        _p(k_cache) if store else None, _p(v_cache) if store else None,
        _p(slot_mapping) if store else None, n_slots, T, nh, nkv, hd, _stream()),
        "glm_qkv_split_norm_rope_store")
    return q, k, v


def gemm_qkv_rope_store(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None, nh: int, nkv: int, hd: int,
                        positions: torch.Tensor, cos_sin_cache: torch.Tensor, rotary_dim: int,
                        k_cache: torch.Tensor | None, v_cache: torch.Tensor | None, slot_mapping: torch.Tensor | None,
                        need_kv: bool):
    """Prefill: QKV projection + split + rotary + K/V cache store in ONE launch (glm_gemm_qkv_rope_store).  Returns
    (q [T,nh,hd], k, v [T,nkv,hd] or None) — or None when the launcher does not cover the shape (nothing was enqueued:
    the caller takes gemm_bf16 + qkv_split_norm_rope_store)."""
    if x.dim() != 2 or x.shape[0] < 96 or hd != 128 or rotary_dim != 64 or opt("fuse_qkv_rope") == 0:
        return None
    _bf16(x, "x"); _bf16(w, "w")
    T, K = x.shape
    assert w.shape == ((nh + 2 * nkv) * hd, K) and w.stride(1) == 1 and positions.dtype == torch.int64
    assert cos_sin_cache.dtype == torch.float32 and cos_sin_cache.is_contiguous() and cos_sin_cache.shape[-1] == rotary_dim
    q = torch.empty((T, nh, hd), dtype=x.dtype, device=x.device)
    k = torch.empty((T, nkv, hd), dtype=x.dtype, device=x.device) if need_kv else None
    v = torch.empty_like(k) if need_kv else None
    has_cache = k_cache is not None and slot_mapping is not None
    if has_cache and slot_mapping.numel() != T:
        # never skip the K/V store silently: later decode steps would read unwritten cache rows (kv_store asserts too)
        raise ValueError(f"gemm_qkv_rope_store: slot_mapping has {slot_mapping.numel()} entries for {T} token rows")
    if positions.numel() != T or positions.device != x.device or (bias is not None and bias.data_ptr() % 8):
        return None   # the two-launch path checks / handles these
    if has_cache:
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and slot_mapping.dtype == torch.int32
    n_slots = k_cache.shape[0] * k_cache.shape[1] if has_cache else 0
    rc = load_library().glm_gemm_qkv_rope_store(
        _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), T, nh, nkv, hd, K, _p(positions), _p(cos_sin_cache), rotary_dim,
        _p(q), _p(k), _p(v), _p(k_cache) if has_cache else None, _p(v_cache) if has_cache else None,
        _p(slot_mapping) if has_cache else None, n_slots, _stream())
    if rc == GLM_ERR_UNSUPPORTED:
        return None
    _check(rc, "glm_gemm_qkv_rope_store")
    return q, k, v


def padded_weight(rows: int, cols: int, *lead: int, dtype=None, device=None) -> torch.Tensor:
    """Allocate [*lead, rows, cols] whose row pitch avoids HBM channel aliasing.

    A wave of the streaming GEMM reads 16-32 weight rows at once; when the row pitch is a multiple
    of 4 KiB those rows land on the same channels (measured 4.7 vs 5.8 TB/s at K=4096).  64 extra
    elements per row (never read) fix that; the returned tensor is the [.., rows, cols] view."""
    pad = 64 if (cols * 2) % 4096 == 0 else 0
    t = torch.empty((*lead, rows, cols + pad), dtype=dtype, device=device)
    return t[..., :cols] if pad else t


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    _bf16(x, "x")
    I = x.shape[-1] // 2
    x2 = x.reshape(-1, 2 * I).contiguous()
    y = torch.empty((x2.shape[0], I), dtype=x.dtype, device=x.device)
    _check(load_library().glm_silu_and_mul(_p(x2), _p(y), x2.shape[0], I, _stream()),
           "glm_silu_and_mul")
    return y.view(*x.shape[:-1], I)


# ------------------------------------------------------------------------------------------
# attention
# Split-KV scratch of the decode attention: ONE grow-only buffer per device, sized for the split count the launch
# really uses (none at all once B * nkv >= 256).  Keyed by device only: launches of one engine on one device are stream
# ordered (eager step, then capture on torch's side stream, then replays), and a per-stream key would make every graph
# capture miss the buffer reserve_attn_workspace() laid down on the default stream.  A replaced buffer stays referenced:
# HIP graphs captured earlier hold its address.  reserve_attn_workspace() lets the runner size it for max_num_seqs
# BEFORE the KV cache takes the rest of the memory; growing inside a stream capture is an error (the allocation and
# its memset would land in that graph's private pool).
_attn_ws: dict = {}
_attn_ws_retired: list = []


def _grow_only(table: dict, retired: list, device, need: int, what: str) -> torch.Tensor:
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    ws = table.get(key)
    if ws is None or ws.numel() < need:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"{what}: {need} bytes needed while a HIP graph is being captured, {0 if ws is None else ws.numel()} "
                               "reserved — run one eager step (or the reserve_* call) at this size before capturing")
        if ws is not None:
            retired.append(ws)
        ws = torch.zeros(need, dtype=torch.uint8, device=device)  # zeros: the head of it holds arrival counters
        table[key] = ws
    return ws


def _attn_workspace(device, need: int) -> torch.Tensor | None:
    if need <= 0:
        return None
    return _grow_only(_attn_ws, _attn_ws_retired, device, need, "decode attention split-KV scratch")


def attn_workspace_bytes(B: int, nh: int, nkv: int, hd: int, block_size: int, bt_stride: int, num_splits: int = 0,
                         max_context_len: int = 0) -> int:
    lib = load_library()
    s = lib.glm_paged_attn_decode_splits(B, nkv, block_size, bt_stride, num_splits, max_context_len)
    return lib.glm_paged_attn_decode_workspace_bytes(B, nh, hd, s) if s > 1 else 0


def reserve_attn_workspace(device, max_batch: int, nh: int, nkv: int, hd: int, block_size: int, bt_stride: int) -> int:
    """Size the split-KV scratch for every decode batch size up to max_batch (worst case over B); returns the bytes."""
    need = max((attn_workspace_bytes(b, nh, nkv, hd, block_size, bt_stride) for b in range(1, max_batch + 1)), default=0)
    _attn_workspace(device, need)
    return need


def paged_attn_decode(q: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                      context_lens: torch.Tensor, block_tables: torch.Tensor, scale: float,
                      num_splits: int = 0, max_context_len: int = 0) -> torch.Tensor:
    """q: [B,nh,hd]; caches [n_blocks, block_size, nkv, hd] -> [B,nh,hd]."""
    _bf16(q, "q"); _bf16(k_cache, "k_cache"); _bf16(v_cache, "v_cache")
    B, nh, hd = q.shape
    n_blocks, block_size, nkv, _ = k_cache.shape
    assert q.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert context_lens.dtype == torch.int32 and block_tables.dtype == torch.int32
    assert block_tables.stride(1) == 1
    lib = load_library()
    need = attn_workspace_bytes(B, nh, nkv, hd, block_size, block_tables.stride(0), num_splits, max_context_len)
    ws = _attn_workspace(q.device, need)
    out = torch.empty_like(q)
    _check(lib.glm_paged_attn_decode(_p(q), _p(k_cache), _p(v_cache), _p(context_lens),
                                     _p(block_tables), block_tables.stride(0), _p(out), B, nh, nkv,
                                     hd, block_size, n_blocks, scale, num_splits, max_context_len,
                                     _p(ws), ws.numel() if ws is not None else 0, _stream()), "glm_paged_attn_decode")
    return out


def paged_attn_decode_qkv(qkv, nh: int, nkv: int, hd: int, positions: torch.Tensor,
                          cos_sin_cache: torch.Tensor, rotary_dim: int, slot_mapping: torch.Tensor,
                          k_cache: torch.Tensor, v_cache: torch.Tensor, context_lens: torch.Tensor,
                          block_tables: torch.Tensor, scale: float, num_splits: int = 0,
                          max_context_len: int = 0, qkv_bias: torch.Tensor | None = None) -> torch.Tensor:
    """Decode attention straight from the packed QKV rows [B, (nh+2nkv)*hd]: rotary on Q/K, K/V cache store and the
    paged attention in one launch (no head RMSNorm, rotary_dim 64).  qkv may be a PartialSum (K-slice partial sums of
    the QKV projection, gemm_bf16_partial) together with the projection's bias: the kernel reduces them on load."""
    _bf16(k_cache, "k_cache"); _bf16(v_cache, "v_cache")
    partial = isinstance(qkv, PartialSum)
    if partial:
        part = qkv.part
        assert part.is_contiguous() and part.dtype == torch.float32 and part.shape[2] == (nh + 2 * nkv) * hd
        if qkv_bias is not None:
            _bf16(qkv_bias, "qkv_bias")
            assert qkv_bias.is_contiguous() and qkv_bias.numel() == part.shape[2]
        B = part.shape[1]
        dev_, dt_ = part.device, k_cache.dtype
    else:
        _bf16(qkv, "qkv")
        assert qkv_bias is None, "a bias belongs to partial sums only (bf16 rows already include it)"
        B = qkv.shape[0]
        dev_, dt_ = qkv.device, qkv.dtype
    n_blocks, block_size, nkv_c, _ = k_cache.shape
    assert nkv_c == nkv and (partial or (qkv.dim() == 2 and qkv.stride(1) == 1))
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and cos_sin_cache.dtype == torch.float32
    assert cos_sin_cache.is_contiguous() and cos_sin_cache.shape[-1] == rotary_dim
    assert positions.dtype == torch.int64 and slot_mapping.dtype == torch.int32
    assert context_lens.dtype == torch.int32 and block_tables.dtype == torch.int32 and block_tables.stride(1) == 1
    lib = load_library()
    need = attn_workspace_bytes(B, nh, nkv, hd, block_size, block_tables.stride(0), num_splits, max_context_len)
    ws = _attn_workspace(dev_, need)
    out = torch.empty((B, nh, hd), dtype=dt_, device=dev_)
    if partial:
        _check(lib.glm_paged_attn_decode_qkv_partial(
            _p(part), part.shape[0], part.shape[1] * part.shape[2], _p(qkv_bias), _p(positions), _p(cos_sin_cache),
            cos_sin_cache.numel() // rotary_dim, rotary_dim, _p(slot_mapping), _p(k_cache), _p(v_cache), _p(context_lens),
            _p(block_tables), block_tables.stride(0), _p(out), B, nh, nkv, hd, block_size, n_blocks, scale, num_splits,
            max_context_len, _p(ws), ws.numel() if ws is not None else 0, _stream()), "glm_paged_attn_decode_qkv_partial")
        return out
    _check(lib.glm_paged_attn_decode_qkv(_p(qkv), qkv.stride(0), _p(positions), _p(cos_sin_cache),
                                         cos_sin_cache.numel() // rotary_dim, rotary_dim,
                                         _p(slot_mapping), _p(k_cache), _p(v_cache), _p(context_lens),
                                         _p(block_tables), block_tables.stride(0), _p(out), B, nh, nkv, hd,
                                         block_size, n_blocks, scale, num_splits, max_context_len, _p(ws),
                                         ws.numel() if ws is not None else 0, _stream()), "glm_paged_attn_decode_qkv")
    return out


def varlen_attn_prefill(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, scale,
                        block_tables=None, block_size=256) -> torch.Tensor:
    _bf16(q, "q"); _bf16(k, "k"); _bf16(v, "v")
    Tq, nh, hd = q.shape
    nkv = k.shape[-2]
    B = cu_seqlens_q.numel() - 1
    out = torch.empty((Tq, nh, hd), dtype=q.dtype, device=q.device)
    q = q.contiguous(); k = k.contiguous(); v = v.contiguous()
    bts = block_tables.stride(0) if block_tables is not None else 0
    _check(load_library().glm_varlen_attn_prefill(_p(q), _p(k), _p(v), _p(cu_seqlens_q),
                                                  _p(cu_seqlens_k), _p(block_tables), bts, _p(out),
                                                  B, max_seqlen_q, nh, nkv, hd, block_size, scale,
                                                  _stream()), "glm_varlen_attn_prefill")
    return out


# ------------------------------------------------------------------------------------------
# MoE
# From this many rows on the router's logits (R1, moe.py:33: bf16(x . Wg^T)) are computed as what they are at that size,
# a [T, H] x [H, E] GEMM, and glm_router_select does R2..R7 from them; below, the token-per-workgroup kernel
# (glm_router_topk / the fused add-RMSNorm + router + bucketing launch), which is built for decode latency: at T = 16384
# it takes 1.2 ms per layer against 0.34 ms for norm + GEMM + select + bucketing (tools/prefill_router_probe.py).
ROUTER_GEMM_MIN_ROWS = 1024


def router_by_gemm(T: int, E: int) -> bool:
    return T >= ROUTER_GEMM_MIN_ROWS and E % 8 == 0  # the GEMM kernels want 16-byte output rows


def router_select(logits: torch.Tensor, bias: torch.Tensor, k: int, routed_scaling_factor: float):
    """R2..R7 from bf16 logits [T, E] (glm_router_select): (topk_idx int64 ascending per row, topk_w bf16)."""
    _bf16(logits, "logits")
    assert bias.dtype == torch.float32 and logits.dim() == 2 and logits.stride(1) == 1
    T, E = logits.shape
    idx = torch.empty((T, k), dtype=torch.int64, device=logits.device)
    w = torch.empty((T, k), dtype=torch.bfloat16, device=logits.device)
    _check(load_library().glm_router_select(_p(logits), logits.stride(0), _p(bias), _p(idx), _p(w), T, E, k,
                                            routed_scaling_factor, _stream()), "glm_router_select")
    return idx, w


def router_topk(x: torch.Tensor, gate_w: torch.Tensor, bias: torch.Tensor, k: int,
                routed_scaling_factor: float, return_logits: bool = False):
    _bf16(x, "x"); _bf16(gate_w, "gate_w")
    assert bias.dtype == torch.float32
    T, H = x.shape
    E = gate_w.shape[0]
    x = x.contiguous()
    if router_by_gemm(T, E):
        logits = gemm_bf16(x, gate_w, None)
        idx, w = router_select(logits, bias, k, routed_scaling_factor)
        return (idx, w, logits) if return_logits else (idx, w)
    idx = torch.empty((T, k), dtype=torch.int64, device=x.device)
    w = torch.empty((T, k), dtype=torch.bfloat16, device=x.device)
    logits = torch.empty((T, E), dtype=torch.bfloat16, device=x.device) if return_logits else None
    _check(load_library().glm_router_topk(_p(x), _p(gate_w), _p(bias), _p(idx), _p(w), _p(logits),
                                          T, H, E, k, routed_scaling_factor, _stream()),
           "glm_router_topk")
    return (idx, w, logits) if return_logits else (idx, w)


_router_ws: dict = {}
_router_ws_retired: list = []


def _router_workspace(device, T: int, E: int) -> torch.Tensor:
    """Workspace of the fused router + align launch (arrival counters, zero-initialised and restored by the kernel, then
    the T x E logits scratch): one grow-only buffer per device (see _grow_only); a replaced buffer stays referenced
    because HIP graphs captured earlier hold its address."""
    return _grow_only(_router_ws, _router_ws_retired, device, load_library().glm_router_workspace_bytes(T, E),
                      "router workspace")


def reserve_router_workspace(device, max_tokens: int, E: int) -> int:
    """Size the router workspace for every batch size up to max_tokens (the runner calls this before graph capture)."""
    need = load_library().glm_router_workspace_bytes(max_tokens, E)
    _grow_only(_router_ws, _router_ws_retired, device, need, "router workspace")
    return need


def add_rmsnorm_router_topk(x, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float,
                            gate_w: torch.Tensor, bias: torch.Tensor, k: int, routed_scaling_factor: float,
                            align_with_shared: bool | None = None):
    """Post-attention add-RMSNorm + router in one launch.  x: bf16 [T,H] tensor or PartialSum (o_proj partial sums).
    Returns (normalised hidden, residual_out, topk_idx, topk_w); with align_with_shared = True/False the same launch
    also builds moe_align's (expert_offsets, sorted_token, sorted_scale, pair_pos), returned as a fifth element."""
    _bf16(residual, "residual"); _bf16(norm_weight, "norm_weight"); _bf16(gate_w, "gate_w")
    assert bias.dtype == torch.float32
    H = residual.shape[-1]
    r2 = residual.reshape(-1, H).contiguous()
    T = r2.shape[0]
    E = gate_w.shape[0]
    if router_by_gemm(T, E) and not isinstance(x, PartialSum):
        # prefill sizes: norm, logits GEMM + selection, bucketing as separate launches (see ROUTER_GEMM_MIN_ROWS)
        yn, rn = add_rmsnorm(x.reshape(-1, H), r2, norm_weight, eps)
        idx, w = router_topk(yn, gate_w, bias, k, routed_scaling_factor)
        if align_with_shared is None:
            return yn.view(residual.shape), rn.view(residual.shape), idx, w
        return (yn.view(residual.shape), rn.view(residual.shape), idx, w,
                moe_align(idx, w, E, bool(align_with_shared)))
    y = torch.empty_like(r2)
    ro = torch.empty_like(r2)
    idx = torch.empty((T, k), dtype=torch.int64, device=r2.device)
    w = torch.empty((T, k), dtype=torch.bfloat16, device=r2.device)
    if isinstance(x, PartialSum):
        part = x.part
        assert part.is_contiguous() and part.shape[1:] == (T, H)
        xp, pp, ns, stride = None, _p(part), part.shape[0], T * H
    else:
        _bf16(x, "x")
        x2 = x.reshape(-1, H).contiguous()
        assert x2.shape[0] == T
        xp, pp, ns, stride = _p(x2), None, 0, 0
    if align_with_shared is not None and 8 * (E + 1) <= 2 * H:
        ws = bool(align_with_shared)
        G, R, dev = E + (1 if ws else 0), T * k + (T if ws else 0), r2.device
        offs = torch.empty(G + 1, dtype=torch.int32, device=dev)
        tok = torch.empty(R, dtype=torch.int32, device=dev)
        scale = torch.empty(R, dtype=torch.bfloat16, device=dev)
        pos = torch.empty(T * k, dtype=torch.int32, device=dev)
        rws = _router_workspace(dev, T, E)
        _check(load_library().glm_add_rmsnorm_router_topk_align(
            xp, pp, ns, stride, _p(r2), _p(norm_weight), eps, _p(y), _p(ro), _p(gate_w), _p(bias), _p(idx), _p(w), T, H,
            E, k, routed_scaling_factor, _p(offs), _p(tok), _p(scale), _p(pos), 1 if ws else 0, _p(rws), rws.numel(),
            _stream()), "glm_add_rmsnorm_router_topk_align")
        return y.view(residual.shape), ro.view(residual.shape), idx, w, (offs, tok, scale, pos)
    _check(load_library().glm_add_rmsnorm_router_topk(xp, pp, ns, stride, _p(r2), _p(norm_weight), eps, _p(y), _p(ro),
                                                      _p(gate_w), _p(bias), _p(idx), _p(w), T, H, E, k,
                                                      routed_scaling_factor, _stream()),
           "glm_add_rmsnorm_router_topk")
    if align_with_shared is not None:
        return y.view(residual.shape), ro.view(residual.shape), idx, w, moe_align(idx, w, E, bool(align_with_shared))
    return y.view(residual.shape), ro.view(residual.shape), idx, w


ALIGN_PARALLEL_MIN_PAIRS = 16384  # from here on the one-workgroup bucketing kernel is the slower one (200 us at 131072 pairs)


def moe_align(topk_idx: torch.Tensor, topk_w: torch.Tensor, E: int, with_shared: bool):
    T, k = topk_idx.shape
    G = E + (1 if with_shared else 0)
    R = T * k + (T if with_shared else 0)
    dev = topk_idx.device
    offs = torch.empty(G + 1, dtype=torch.int32, device=dev)
    tok = torch.empty(R, dtype=torch.int32, device=dev)
    scale = torch.empty(R, dtype=torch.bfloat16, device=dev)
    pos = torch.empty(T * k, dtype=torch.int32, device=dev)
    if T * k >= ALIGN_PARALLEL_MIN_PAIRS:  # prefill sizes: count / offsets / scatter over many workgroups
        lib = load_library()
        ws = torch.zeros(lib.glm_moe_align_workspace_bytes(T, k, E) // 4, dtype=torch.int32, device=dev)
        _check(lib.glm_moe_align_parallel(_p(topk_idx), _p(topk_w), _p(offs), _p(tok), _p(scale), _p(pos), T, k, E,
                                          1 if with_shared else 0, _p(ws), ws.numel() * 4, _stream()),
               "glm_moe_align_parallel")
        return offs, tok, scale, pos
    _check(load_library().glm_moe_align(_p(topk_idx), _p(topk_w), _p(offs), _p(tok), _p(scale),
                                        _p(pos), T, k, E, 1 if with_shared else 0, _stream()),
           "glm_moe_align")
    return offs, tok, scale, pos


TINYM_MAX_ROWS = 1  # token rows the single-sequence GEMV body takes (csrc/gemm_tinym.hip)


def grouped_gemm_gateup_silu(x, w13, expert_offsets, sorted_token, R: int, max_rows_per_group: int | None = None) -> torch.Tensor:
    """x [T,H]; w13 [G, 2I, H] -> act [R, I].  max_rows_per_group: an upper bound the caller knows (T for a top-k routing
    with the shared group); with ONE token row it selects the single-sequence GEMV body."""
    _bf16(x, "x"); _bf16(w13, "w13")
    G, I2, H = w13.shape
    I = I2 // 2
    assert w13.stride(2) == 1  # rows may be padded (stride(1) >= H), groups strided
    act = torch.empty((R, I), dtype=x.dtype, device=x.device)
    if x.shape[0] <= TINYM_MAX_ROWS and max_rows_per_group is not None and max_rows_per_group <= x.shape[0]:
        # single-sequence decode: the GEMV body (the caller vouches that no group holds more rows than there are tokens;
        # the body computes ONE row per group and would silently drop further ones — at least the row count must allow it)
        if R > G:
            raise ValueError(f"grouped_gemm_gateup_silu: {R} rows over {G} groups cannot be one row per group")
        _check(load_library().glm_grouped_gemm_gateup_silu_fewtok(
            _p(x), x.stride(0), x.shape[0], _p(w13), w13.stride(0), w13.stride(1), _p(expert_offsets), _p(sorted_token),
            _p(act), R, H, I, G, _stream()), "glm_grouped_gemm_gateup_silu_fewtok")
        return act
    _check(load_library().glm_grouped_gemm_gateup_silu(_p(x), x.stride(0), _p(w13), w13.stride(0),
                                                       w13.stride(1),
                                                       _p(expert_offsets), _p(sorted_token),
                                                       _p(act), R, H, I, G, _stream()),
           "glm_grouped_gemm_gateup_silu")
    return act


def grouped_gemm_down(act, w2, expert_offsets, sorted_scale, n_scaled_groups: int,
                      max_rows_per_group: int | None = None) -> torch.Tensor:
    """act [R,I]; w2 [G, H, I] -> y [R, H]."""
    _bf16(act, "act"); _bf16(w2, "w2")
    G, H, I = w2.shape
    R = act.shape[0]
    assert w2.stride(2) == 1 and act.is_contiguous()
    y = torch.empty((R, H), dtype=act.dtype, device=act.device)
    if max_rows_per_group is not None and max_rows_per_group <= TINYM_MAX_ROWS:
        if act.shape[0] > w2.shape[0] * max(1, max_rows_per_group):
            raise ValueError(f"grouped_gemm_down: {act.shape[0]} rows over {w2.shape[0]} groups exceed {max_rows_per_group} per group")
        _check(load_library().glm_grouped_gemm_down_fewtok(_p(act), max_rows_per_group, _p(w2), w2.stride(0), w2.stride(1),
                                                           _p(expert_offsets), _p(sorted_scale), n_scaled_groups, _p(y), H,
                                                           R, H, I, G, _stream()), "glm_grouped_gemm_down_fewtok")
        return y
    _check(load_library().glm_grouped_gemm_down(_p(act), _p(w2), w2.stride(0), w2.stride(1),
                                                _p(expert_offsets),
                                                _p(sorted_scale), n_scaled_groups, _p(y), H, R, H,
                                                I, G, _stream()), "glm_grouped_gemm_down")
    return y


FP8_MAX = 448.0  # largest finite OCP e4m3fn value


def quantize_fp8_rowwise(w: torch.Tensor):
    """Weight-only quantisation used for the experts: per-row scale s = max|row| / 448,
    q = e4m3fn(row / s).  Returns (q as uint8 with a padded row pitch, s as fp32)."""
    rows, cols = w.shape[-2], w.shape[-1]
    s = (w.abs().amax(dim=-1).to(torch.float32) / FP8_MAX).clamp_min(1e-12)
    q = (w.to(torch.float32) / s.unsqueeze(-1)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    pad = 128 if cols % 4096 == 0 else 0  # same channel-aliasing rule as padded_weight (bytes)
    store = torch.empty((*w.shape[:-1], cols + pad), dtype=torch.uint8, device=w.device)
    view = store[..., :cols]
    view.copy_(q.view(torch.uint8))
    return view, s.contiguous()


def quantize_fp8_rows(x: torch.Tensor):
    """Dynamic per-row e4m3 quantisation of activation rows (glm_quantize_fp8_rows): x [T, H] bf16 ->
    (q [T, H] uint8, scale [T] fp32) with x ~ q * scale[:, None]."""
    _bf16(x, "x")
    assert x.dim() == 2 and x.stride(1) == 1
    T, H = x.shape
    q = torch.empty((T, H), dtype=torch.uint8, device=x.device)
    s = torch.empty(T, dtype=torch.float32, device=x.device)
    _check(load_library().glm_quantize_fp8_rows(_p(x), x.stride(0), _p(q), q.stride(0), _p(s), T, H, _stream()),
           "glm_quantize_fp8_rows")
    return q, s


def grouped_gemm_gateup_silu_fp8a8(x_q, x_scale, w13_q, w13_s, expert_offsets, sorted_token, R: int):
    """fp8 x fp8 gate|up on the block-scaled MFMA (glm_grouped_gemm_gateup_silu_fp8a8).  x_q [T, H] uint8 + x_scale [T];
    w13_q [G, 2I, H] uint8 + w13_s [G, 2I].  Returns (act_q [R, I] uint8, act_bscale [R, 64] uint8: E8M0 per 32 columns) — or
    None when the launcher does not cover the shape (nothing enqueued: take the weight-only path)."""
    G, I2, H = w13_q.shape
    I = I2 // 2
    assert x_q.dtype == torch.uint8 and w13_q.dtype == torch.uint8 and x_q.stride(1) == 1 and w13_q.stride(2) == 1
    assert x_scale.dtype == torch.float32 and w13_s.dtype == torch.float32 and w13_s.stride(1) == 1
    act = torch.empty((R, I), dtype=torch.uint8, device=x_q.device)
    bs = torch.zeros((R, 64), dtype=torch.uint8, device=x_q.device)
    rc = load_library().glm_grouped_gemm_gateup_silu_fp8a8(
        _p(x_q), x_q.stride(0), _p(x_scale), _p(w13_q), w13_q.stride(0), w13_q.stride(1), _p(w13_s), w13_s.stride(0),
        _p(expert_offsets), _p(sorted_token), _p(act), act.stride(0), _p(bs), R, H, I, G, _stream())
    if rc == GLM_ERR_UNSUPPORTED:
        return None
    _check(rc, "glm_grouped_gemm_gateup_silu_fp8a8")
    return act, bs


def grouped_gemm_down_fp8a8(act_q, act_bscale, w2_q, w2_s, expert_offsets, sorted_scale, n_scaled_groups: int):
    """fp8 x fp8 down projection with the act rows' E8M0 block scales inside the MFMA; y [R, H] bf16, or None (shape not
    covered)."""
    G, H, I = w2_q.shape
    R = act_q.shape[0]
    assert act_q.dtype == torch.uint8 and w2_q.dtype == torch.uint8 and act_bscale.dtype == torch.uint8
    assert act_bscale.shape == (R, 64) and act_bscale.is_contiguous() and w2_q.stride(2) == 1 and w2_s.stride(1) == 1
    y = torch.empty((R, H), dtype=torch.bfloat16, device=act_q.device)
    rc = load_library().glm_grouped_gemm_down_fp8a8(
        _p(act_q), act_q.stride(0), _p(act_bscale), _p(w2_q), w2_q.stride(0), w2_q.stride(1), _p(w2_s), w2_s.stride(0),
        _p(expert_offsets), _p(sorted_scale), n_scaled_groups, _p(y), y.stride(0), R, H, I, G, _stream())
    if rc == GLM_ERR_UNSUPPORTED:
        return None
    _check(rc, "glm_grouped_gemm_down_fp8a8")
    return y


_FP8_LARGE_MIN_AVG_ROWS = 96  # api.hip LARGEM_MIN_AVG_ROWS
_fp8_scratch: dict = {}


def _fp8_expand_scratch(device, G: int, rows: int, cols: int, R: int):
    """Scratch for the prefill-regime fp8 expert GEMM where the tiled kernel cannot take the fp8 bytes directly (K not a
    multiple of 64): bf16 copy of ONE weight stack, reused by every layer and both projections.  None at decode row counts
    (streaming fp8 kernel) and for the shapes the fp8-operand kernel covers."""
    if R // max(G, 1) < _FP8_LARGE_MIN_AVG_ROWS:
        return None, 0
    need = load_library().glm_fp8w_expand_scratch_bytes(G, rows, cols)
    if need == 0:  # the tiled kernel takes the fp8 bytes themselves
        return None, 0
    buf = _fp8_scratch.get(device)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _fp8_scratch[device] = buf
    return buf, buf.numel()


def grouped_gemm_gateup_silu_fp8w(x, w13_q, w13_s, expert_offsets, sorted_token, R: int) -> torch.Tensor:
    """x [T,H] bf16; w13_q [G,2I,H] uint8 (fp8 e4m3); w13_s [G,2I] fp32 -> act [R,I]."""
    _bf16(x, "x")
    G, I2, H = w13_q.shape
    I = I2 // 2
    assert w13_q.dtype == torch.uint8 and w13_q.stride(2) == 1 and w13_s.dtype == torch.float32
    assert w13_s.stride(1) == 1
    act = torch.empty((R, I), dtype=x.dtype, device=x.device)
    scr, scr_n = _fp8_expand_scratch(x.device, G, I2, H, R)
    _check(load_library().glm_grouped_gemm_gateup_silu_fp8w(
        _p(x), x.stride(0), _p(w13_q), w13_q.stride(0), w13_q.stride(1), _p(w13_s), w13_s.stride(0),
        _p(expert_offsets), _p(sorted_token), _p(act), R, H, I, G, _p(scr), scr_n, _stream()),
        "glm_grouped_gemm_gateup_silu_fp8w")
    return act


def grouped_gemm_down_fp8w(act, w2_q, w2_s, expert_offsets, sorted_scale, n_scaled_groups: int) -> torch.Tensor:
    _bf16(act, "act")
    G, H, I = w2_q.shape
    R = act.shape[0]
    assert w2_q.dtype == torch.uint8 and w2_q.stride(2) == 1 and w2_s.dtype == torch.float32 and act.is_contiguous()
    y = torch.empty((R, H), dtype=act.dtype, device=act.device)
    scr, scr_n = _fp8_expand_scratch(act.device, G, H, I, R)
    _check(load_library().glm_grouped_gemm_down_fp8w(
        _p(act), _p(w2_q), w2_q.stride(0), w2_q.stride(1), _p(w2_s), w2_s.stride(0), _p(expert_offsets),
        _p(sorted_scale), n_scaled_groups, _p(y), H, R, H, I, G, _p(scr), scr_n, _stream()),
        "glm_grouped_gemm_down_fp8w")
    return y


def ep_plan(topk_idx: torch.Tensor, topk_w: torch.Tensor, E: int, ep: int, rank: int, cap: int):
    """Fixed-capacity expert-parallel dispatch plan (glm_ep_plan) from the replicated routing [T, k]: device tensors
    (send_gather, slot_scale, pair_slot, recv_offs, recv_gidx, overflow); no host synchronisation."""
    T, k = topk_idx.shape
    _bf16(topk_w, "topk_w")
    assert topk_idx.dtype == torch.int64 and topk_idx.is_contiguous() and topk_w.is_contiguous() and E % ep == 0
    dev = topk_idx.device
    per = (T + ep - 1) // ep
    Tl = max(0, min(T, (rank + 1) * per) - min(T, rank * per))
    send_gather = torch.empty(ep * cap, dtype=torch.int32, device=dev)
    slot_scale = torch.zeros(ep * cap, dtype=torch.bfloat16, device=dev)
    pair_slot = torch.empty(max(Tl * k, 1), dtype=torch.int32, device=dev)
    recv_offs = torch.empty(E // ep + 1, dtype=torch.int32, device=dev)
    recv_gidx = torch.empty(ep * cap, dtype=torch.int32, device=dev)
    overflow = torch.empty(1, dtype=torch.int32, device=dev)
    _check(load_library().glm_ep_plan(_p(topk_idx), _p(topk_w), T, k, E, ep, rank, cap, _p(send_gather), _p(slot_scale),
                                      _p(pair_slot), _p(recv_offs), _p(recv_gidx), _p(overflow), _stream()), "glm_ep_plan")
    return send_gather, slot_scale, pair_slot, recv_offs, recv_gidx, overflow


def grouped_gemm_gateup_silu_cap(x, w13, expert_offsets, sorted_token, R_cap: int, R_expected: int) -> torch.Tensor:
    """grouped_gemm_gateup_silu over a capacity-sized row array: expert_offsets say which of the R_cap rows are real."""
    _bf16(x, "x"); _bf16(w13, "w13")
    G, I2, H = w13.shape
    I = I2 // 2
    assert w13.stride(2) == 1
    act = torch.empty((R_cap, I), dtype=x.dtype, device=x.device)
    _check(load_library().glm_grouped_gemm_gateup_silu_cap(_p(x), x.stride(0), _p(w13), w13.stride(0), w13.stride(1),
                                                           _p(expert_offsets), _p(sorted_token), _p(act), R_cap, R_expected,
                                                           H, I, G, _stream()), "glm_grouped_gemm_gateup_silu_cap")
    return act


def grouped_gemm_down_cap(act, w2, expert_offsets, R_expected: int) -> torch.Tensor:
    """Unscaled grouped down projection over a capacity-sized row array (the routing weight is applied at home)."""
    _bf16(act, "act"); _bf16(w2, "w2")
    G, H, I = w2.shape
    R_cap = act.shape[0]
    assert w2.stride(2) == 1 and act.is_contiguous()
    y = torch.empty((R_cap, H), dtype=act.dtype, device=act.device)
    _check(load_library().glm_grouped_gemm_down_cap(_p(act), _p(w2), w2.stride(0), w2.stride(1), _p(expert_offsets), None, 0,
                                                    _p(y), H, R_cap, R_expected, H, I, G, _stream()),
           "glm_grouped_gemm_down_cap")
    return y


def moe_combine(y, pair_pos, T: int, k: int, shared_base: int) -> torch.Tensor:
    H = y.shape[1]
    out = torch.empty((T, H), dtype=y.dtype, device=y.device)
    _check(load_library().glm_moe_combine(_p(y), y.stride(0), _p(pair_pos), _p(out), T, k, H,
                                          shared_base, _stream()), "glm_moe_combine")
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """dst[i] = src[idx[i]] (rows of bf16; idx int32; idx < 0: the row is left uninitialised)."""
    _bf16(src, "src")
    assert src.dim() == 2 and idx.dtype == torch.int32 and idx.is_contiguous()
    dst = torch.empty((idx.numel(), src.shape[1]), dtype=src.dtype, device=src.device)
    _check(load_library().glm_gather_rows(_p(src), src.stride(0), _p(dst), dst.stride(0), _p(idx), idx.numel(),
                                          src.shape[1], _stream()), "glm_gather_rows")
    return dst


def scatter_rows(src: torch.Tensor, idx: torch.Tensor, n_rows: int | None = None, no_init: bool = False) -> torch.Tensor:
    """dst[idx[i]] = src[i] (idx < 0: skipped); rows of dst no index points at stay zero (no_init: uninitialised)."""
    _bf16(src, "src")
    assert src.dim() == 2 and idx.dtype == torch.int32 and idx.is_contiguous() and idx.numel() == src.shape[0]
    alloc = torch.empty if no_init else torch.zeros
    dst = alloc((n_rows if n_rows is not None else src.shape[0], src.shape[1]), dtype=src.dtype, device=src.device)
    _check(load_library().glm_scatter_rows(_p(src), src.stride(0), _p(dst), dst.stride(0), _p(idx), idx.numel(),
                                           src.shape[1], _stream()), "glm_scatter_rows")
    return dst


def moe_combine_scaled(y, pair_pos, row_scale, shared_rows, T: int, k: int) -> torch.Tensor:
    """Expert-parallel combine: routing weight applied here to the unscaled rows, ascending-expert bf16 adds, + shared."""
    _bf16(y, "y"); _bf16(row_scale, "row_scale")
    H = y.shape[1]
    out = torch.empty((T, H), dtype=y.dtype, device=y.device)
    if shared_rows is not None:
        _bf16(shared_rows, "shared_rows")
    _check(load_library().glm_moe_combine_scaled(_p(y), y.stride(0), _p(pair_pos), _p(row_scale), _p(shared_rows),
                                                 shared_rows.stride(0) if shared_rows is not None else 0, _p(out), T, k,
                                                 H, _stream()), "glm_moe_combine_scaled")
    return out


class DeferredCombine:
    """Rows of the grouped down projection plus the per-token row list: the MoE combine (moe_combine) that has not
    been run yet because the consumer — the next layer's add-RMSNorm — performs it in the same launch."""
    __slots__ = ("y", "pair_pos", "T", "k", "shared_base", "out_shape")

    def __init__(self, y, pair_pos, T, k, shared_base, out_shape):
        self.y, self.pair_pos, self.T, self.k, self.shared_base, self.out_shape = y, pair_pos, T, k, shared_base, out_shape

    @property
    def shape(self):
        return self.out_shape

    def materialize(self) -> torch.Tensor:
        return moe_combine(self.y, self.pair_pos, self.T, self.k, self.shared_base).view(self.out_shape)


def moe_combine_add_rmsnorm(dc: DeferredCombine, residual: torch.Tensor, weight: torch.Tensor,
                            eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    """moe_combine + add_rmsnorm(combined, residual) in one launch."""
    _bf16(residual, "residual"); _bf16(weight, "weight")
    H = dc.y.shape[1]
    r2 = residual.reshape(-1, H).contiguous()
    assert r2.shape[0] == dc.T
    out = torch.empty_like(r2)
    ro = torch.empty_like(r2)
    _check(load_library().glm_moe_combine_add_rmsnorm(_p(dc.y), dc.y.stride(0), _p(dc.pair_pos), dc.T, dc.k, H,
                                                      dc.shared_base, _p(r2), _p(weight), _p(out), _p(ro), eps,
                                                      _stream()), "glm_moe_combine_add_rmsnorm")
    return out.view(residual.shape), ro.view(residual.shape)


def combine_add_rmsnorm_gemm(dc: DeferredCombine, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float,
                             w: torch.Tensor, bias: torch.Tensor | None):
    """ONE token: moe_combine + add_rmsnorm(combined, residual) + (x @ w.T + bias) in ONE launch (glm_gemm_bf16_combine_norm).
    Returns (y [T, N], residual_out [T, H]); the normalised hidden state itself is never written."""
    _bf16(residual, "residual"); _bf16(norm_weight, "norm_weight"); _bf16(w, "w")
    H = dc.y.shape[1]
    N = w.shape[0]
    r2 = residual.reshape(-1, H).contiguous()
    assert r2.shape[0] == dc.T and w.shape[1] == H and w.stride(1) == 1
    y = torch.empty((dc.T, N), dtype=r2.dtype, device=r2.device)
    ro = torch.empty_like(r2)
    _check(load_library().glm_gemm_bf16_combine_norm(_p(dc.y), dc.y.stride(0), _p(dc.pair_pos), dc.T, dc.k, dc.shared_base,
                                                     _p(r2), _p(norm_weight), eps, _p(ro), None, _p(w), w.stride(0),
                                                     _p(bias), _p(y), N, N, H, _stream()), "glm_gemm_bf16_combine_norm")
    return y, ro.view(residual.shape)


def can_combine_norm_gemm(dc, H: int) -> bool:
    return (isinstance(dc, DeferredCombine) and dc.T <= TINYM_MAX_ROWS and H % 8 == 0 and H <= 4096
            and opt("tinym") != 0 and opt("fuse_combine_qkv") != 0)


_tail_ws: dict = {}
_tail_ws_retired: list = []


def check_router_tail(device=None) -> None:
    """Raise if a one-token o_proj + router-tail launch gave up its bounded wait since the last call (its outputs were
    poisoned; word 2 of the tail workspace is the sticky status).  One 4-byte read: call it where the step is synchronised
    anyway (ModelRunner.run).  The workspace is re-zeroed so that the next launch starts from clean counters."""
    for key, ws in list(_tail_ws.items()):
        if device is not None and torch.device(device).index not in (None, key[1]):
            continue
        st = int(ws[8:12].view(torch.int32).item())
        if st:
            torch.cuda.synchronize(torch.device(key[0], key[1]))
            ws.zero_()
            raise RuntimeError(f"one-token router tail: designated workgroup {st - 1} gave up waiting for the projection; "
                               "the outputs of that step are NaN")


def can_gemm_router_tail(x: torch.Tensor, H: int, E: int) -> bool:
    return (x.dim() == 2 and x.shape[0] == 1 and 2048 <= H <= 4096 and H % 8 == 0 and E <= 256 and x.is_contiguous()
            and opt("tinym") != 0 and opt("fuse_oproj_router") != 0)


def gemm_add_rmsnorm_router_topk_align(x: torch.Tensor, w: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor,
                                       eps: float, gate_w: torch.Tensor, bias: torch.Tensor, k: int,
                                       routed_scaling_factor: float, with_shared: bool):
    """ONE token: o_proj + post-attention add-RMSNorm + router + bucketing in one launch (glm_gemm_bf16_router_tail).
    x [1, K], w [H, K].  Returns what add_rmsnorm_router_topk(..., align_with_shared=...) returns — or None when the
    launcher does not cover the shape (nothing was enqueued: take the two-launch path)."""
    _bf16(x, "x"); _bf16(w, "w"); _bf16(residual, "residual"); _bf16(norm_weight, "norm_weight"); _bf16(gate_w, "gate_w")
    assert bias.dtype == torch.float32 and x.shape[0] == 1 and w.stride(1) == 1
    K, H, E = x.shape[1], w.shape[0], gate_w.shape[0]
    dev = x.device
    r2 = residual.reshape(1, H).contiguous()
    o_row = torch.empty((1, H), dtype=x.dtype, device=dev)
    y, ro = torch.empty_like(r2), torch.empty_like(r2)
    idx = torch.empty((1, k), dtype=torch.int64, device=dev)
    tw = torch.empty((1, k), dtype=torch.bfloat16, device=dev)
    G, R = E + (1 if with_shared else 0), k + (1 if with_shared else 0)
    offs = torch.empty(G + 1, dtype=torch.int32, device=dev)
    tok = torch.empty(R, dtype=torch.int32, device=dev)
    scale = torch.empty(R, dtype=torch.bfloat16, device=dev)
    pos = torch.empty(k, dtype=torch.int32, device=dev)
    # two self-restoring counters + E floats of logits scratch, one grow-only zeroed buffer per device
    ticket = _grow_only(_tail_ws, _tail_ws_retired, dev, 16 + 4 * 256, "router-tail workspace").data_ptr()
    rc = load_library().glm_gemm_bf16_router_tail(
        _p(x), _p(w), w.stride(0), K, H, _p(o_row), ticket, _p(r2), _p(norm_weight), eps, _p(y), _p(ro), _p(gate_w),
        _p(bias), E, k, routed_scaling_factor, _p(idx), _p(tw), _p(offs), _p(tok), _p(scale), _p(pos), 1 if with_shared else 0,
        _stream())
    if rc == GLM_ERR_UNSUPPORTED:
        # the launcher's own limits (K window of the tail's LDS scratch, rows per wave vs experts) are checked before anything
        # is enqueued: nothing ran, the caller falls back to o_proj + add_rmsnorm_router_topk (two launches)
        return None
    _check(rc, "glm_gemm_bf16_router_tail")
    return y.view(residual.shape), ro.view(residual.shape), idx, tw, (offs, tok, scale, pos)


_engine_ws: dict = {}
_engine_ws_retired: list = []


def can_decode_mlp_engine(x: torch.Tensor, H: int, E: int, k: int, I: int) -> bool:
    return (opt("t1_engine") != 0 and x.dim() == 2 and x.shape[0] == 1 and H in (2048, 4096) and x.shape[1] % 2048 == 0
            and E <= 256 and k <= 64 and 1024 <= I <= 1536 and I % 8 == 0 and x.is_contiguous())


def decode_mlp_engine(x: torch.Tensor, w_o: torch.Tensor, residual: torch.Tensor, norm_weight: torch.Tensor, eps: float,
                      gate_w: torch.Tensor, bias: torch.Tensor, k: int, routed_scaling_factor: float, w13: torch.Tensor,
                      w2: torch.Tensor, want_act: bool = False):
    """ONE token: o_proj + post-attention add-RMSNorm + router + bucketing + grouped gate|up + SiLU*mul + grouped down x
    routing weight in ONE persistent launch (glm_decode_mlp_engine, csrc/decode_engine.hip).  x [1, K]; w_o [H, K];
    w13 [E+1, 2I, H], w2 [E+1, H, I] (group E = the shared expert).  Returns (y_norm, residual_out, topk_idx, topk_w,
    (offs, tok, scale, pos), y [k+1, H], act or None) — exactly what gemm_add_rmsnorm_router_topk_align +
    grouped_gemm_gateup_silu + grouped_gemm_down return, bit for bit — or None when the launcher does not cover the shape
    (nothing was enqueued)."""
    _bf16(x, "x"); _bf16(w_o, "w_o"); _bf16(residual, "residual"); _bf16(norm_weight, "norm_weight"); _bf16(gate_w, "gate_w")
    _bf16(w13, "w13"); _bf16(w2, "w2")
    assert bias.dtype == torch.float32 and x.shape[0] == 1 and w_o.stride(1) == 1 and w13.stride(2) == 1 and w2.stride(2) == 1
    K, H, E = x.shape[1], w_o.shape[0], gate_w.shape[0]
    I = w2.shape[2]
    assert w13.shape[0] == E + 1 and w13.shape[1] == 2 * I and w13.shape[2] == H and w2.shape[0] == E + 1 and w2.shape[1] == H
    dev = x.device
    r2 = residual.reshape(1, H).contiguous()
    y_norm, ro = torch.empty_like(r2), torch.empty_like(r2)
    idx = torch.empty((1, k), dtype=torch.int64, device=dev)
    tw = torch.empty((1, k), dtype=torch.bfloat16, device=dev)
    offs = torch.empty(E + 2, dtype=torch.int32, device=dev)
    tok = torch.empty(k + 1, dtype=torch.int32, device=dev)
    scale = torch.empty(k + 1, dtype=torch.bfloat16, device=dev)
    pos = torch.empty(k, dtype=torch.int32, device=dev)
    y = torch.empty((k + 1, H), dtype=x.dtype, device=dev)
    act = torch.empty((k + 1, I), dtype=x.dtype, device=dev) if want_act else None
    lib = load_library()
    ws = _grow_only(_engine_ws, _engine_ws_retired, dev, lib.glm_decode_mlp_engine_workspace_bytes(H, E, k, I),
                    "decode engine workspace")
    rc = lib.glm_decode_mlp_engine(
        _p(x), _p(w_o), w_o.stride(0), K, H, _p(r2), _p(norm_weight), eps, _p(y_norm), _p(ro), _p(gate_w), _p(bias), E, k,
        routed_scaling_factor, _p(idx), _p(tw), _p(offs), _p(tok), _p(scale), _p(pos), _p(w13), w13.stride(0), w13.stride(1),
        _p(w2), w2.stride(0), w2.stride(1), I, _p(act), _p(y), y.stride(0), ws.data_ptr(), _stream())
    if rc == GLM_ERR_UNSUPPORTED:
        return None
    _check(rc, "glm_decode_mlp_engine")
    return y_norm.view(residual.shape), ro.view(residual.shape), idx, tw, (offs, tok, scale, pos), y, act


def check_decode_engine(device=None) -> None:
    """Raise if a bounded wait inside a glm_decode_mlp_engine launch expired since the last call (word 1 of its workspace is
    the sticky status; y was NaN-filled).  One 4-byte read; the workspace is re-zeroed."""
    for key, ws in list(_engine_ws.items()):
        if device is not None and torch.device(device).index not in (None, key[1]):
            continue
        st = int(ws[4:8].view(torch.int32).item())
        if st:
            torch.cuda.synchronize(torch.device(key[0], key[1]))
            ws.zero_()
            raise RuntimeError(f"one-token decode engine: a bounded wait expired (code 0x{st & 0xf00:x}, workgroup {st & 0xff}); "
                               "the outputs of that step are NaN")


class PartialSum:
    """fp32 K-slice partial sums [nslices, M, N] of a bias-free linear layer whose reduction (and bf16 rounding) is
    deferred to the consumer kernel (add_rmsnorm_partial)."""
    __slots__ = ("part",)

    def __init__(self, part: torch.Tensor):
        self.part = part

    @property
    def shape(self):
        return self.part.shape[1:]


def gemm_bf16_partial(x: torch.Tensor, w: torch.Tensor, want_slices: int = 8) -> PartialSum:
    """K-slice partial sums of x @ w.T for decode batches (M <= 32); see glm_gemm_bf16_partial."""
    _bf16(x, "x"); _bf16(w, "w")
    K = x.shape[-1]
    N = w.shape[0]
    assert x.dim() == 2 and w.shape[1] == K and w.stride(1) == 1 and x.stride(1) == 1
    M = x.shape[0]
    lib = load_library()
    ns = lib.glm_gemm_bf16_partial_slices(K, want_slices)
    part = torch.empty((ns, M, N), dtype=torch.float32, device=x.device)
    _check(lib.glm_gemm_bf16_partial(_p(x), x.stride(0), _p(w), w.stride(0), _p(part), M, N, K, ns, _stream()),
           "glm_gemm_bf16_partial")
    return PartialSum(part)


def add_rmsnorm_partial(ps: PartialSum, residual: torch.Tensor, weight: torch.Tensor,
                        eps: float) -> tuple[torch.Tensor, torch.Tensor]:
    """add_rmsnorm whose x is bf16(sum of the K-slice partial sums)."""
    _bf16(residual, "residual"); _bf16(weight, "weight")
    part = ps.part
    ns, T, H = part.shape
    r2 = residual.reshape(-1, H).contiguous()
    assert r2.shape[0] == T and part.is_contiguous()
    y = torch.empty_like(r2)
    ro = torch.empty_like(r2)
    _check(load_library().glm_add_rmsnorm_partial(_p(part), ns, T * H, _p(r2), _p(weight), _p(y), _p(ro), T, H, eps,
                                                  _stream()), "glm_add_rmsnorm_partial")
    return y.view(residual.shape), ro.view(residual.shape)


def gemm_bf16(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """y = bf16(x @ w.T + bias); x [..., K], w [N, K]."""
    _bf16(x, "x"); _bf16(w, "w")
    K = x.shape[-1]
    N = w.shape[0]
    assert w.shape[1] == K and w.stride(1) == 1  # row pitch w.stride(0) may be padded
    x2 = x.reshape(-1, K)
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    M = x2.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _check(load_library().glm_gemm_bf16(_p(x2), x2.stride(0), _p(w), w.stride(0), _p(bias), _p(y), N,
                                        M, N, K, _stream()), "glm_gemm_bf16")
    return y.view(*x.shape[:-1], N)
