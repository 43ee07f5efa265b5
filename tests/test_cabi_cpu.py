"""CPU-only checks of the drop-in boundary: the shared library builds for gfx950 without a GPU,
loads, and exports every symbol include/glm4moe_hip.h declares; host-only entry points answer;
the product refuses to run without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def lib_path():
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "glm4moe_build", os.path.join(ROOT, "glm4_moe_nano-vllm_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build_lib(force=False, verbose=False)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "glm4moe_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(glm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/glm4moe_hip.h but not exported"


def test_binding_table_matches_header(lib_path):
    from nanovllm import hip_ops
    assert sorted(hip_ops.EXPORTED_SYMBOLS) == declared_symbols()
    lib = hip_ops.load_library(lib_path)
    assert lib.glm_abi_version() == 10


def test_host_only_entry_points(lib_path):
    from nanovllm import hip_ops
    lib = hip_ops.load_library(lib_path)
    assert lib.glm_paged_attn_decode_workspace_bytes(32, 96, 128, 1) == 0
    assert lib.glm_paged_attn_decode_workspace_bytes(2, 8, 128, 4) == 65536 + 2 * 8 * 4 * 130 * 4
    assert lib.glm_paged_attn_decode_splits(32, 8, 256, 16, 0, 4096) == 1 and lib.glm_paged_attn_decode_splits(1, 8, 256, 4, 0, 1024) == 8
    # argument validation happens before any launch: usable without a GPU
    rc = lib.glm_rmsnorm(None, None, None, 4, 128, 1e-5, None)
    assert rc == -1 and b"null pointer" in lib.glm_last_error()
    rc = lib.glm_gemm_bf16(ctypes.c_void_p(16), 8, ctypes.c_void_p(16), 16, None, ctypes.c_void_p(16), 4, 4, 4, 16, None)
    assert rc == -1 and b"bad shape" in lib.glm_last_error()


def test_fp8_expansion_scratch_only_where_the_tiled_kernel_cannot_take_fp8(lib_path):
    """At prefill row counts the tiled kernel takes the fp8 bytes as its weight operand when K is a multiple of its 64-wide K
    tile: no scratch is asked for (Air dims: K = 4096 and 1408); other K, or option fp8w_native = 0, still size the bf16 copy."""
    from nanovllm import hip_ops
    lib = hip_ops.load_library()
    assert hip_ops.opt("fp8w_native") == 1
    assert lib.glm_fp8w_expand_scratch_bytes(65, 2816, 4096) == 0
    assert lib.glm_fp8w_expand_scratch_bytes(65, 4096, 1408) == 0
    assert lib.glm_fp8w_expand_scratch_bytes(3, 512, 352) >= 3 * 512 * 352 * 2
    with hip_ops.option("fp8w_native", 0):
        assert lib.glm_fp8w_expand_scratch_bytes(65, 2816, 4096) >= 65 * 2816 * 4096 * 2
    assert lib.glm_fp8w_expand_scratch_bytes(65, 2816, 4096) == 0


def test_options_are_read_once_and_listed(lib_path):
    """Launch-path switches: one table per side, filled from the environment once, changed only through set_option; a
    later change of the environment is NOT seen (a HIP graph and the eager launches around it cannot disagree)."""
    import os
    from nanovllm import hip_ops
    o = hip_ops.options()
    assert set(o) == {"tinym", "fp8w_native", "largem_variant", "largem_cfg", "fa_mfma32", "fa_rows64", "fa_persist", "fuse_combine_qkv",
                      "fuse_oproj_router", "t1_engine", "defer_qkv_reduce", "fuse_qkv_rope", "ep_fixed", "oneshot_allreduce"}
    assert o["tinym"] == 1 and o["largem_variant"] == -1 and o["oneshot_allreduce"] == 0
    os.environ["GLM4MOE_TINYM"] = "0"
    os.environ["GLM4MOE_FUSE_COMBINE_QKV"] = "0"
    try:
        assert hip_ops.opt("tinym") == 1 and hip_ops.opt("fuse_combine_qkv") == 1
    finally:
        del os.environ["GLM4MOE_TINYM"], os.environ["GLM4MOE_FUSE_COMBINE_QKV"]
    with hip_ops.option("tinym", 0):
        assert hip_ops.options()["tinym"] == 0
    assert hip_ops.opt("tinym") == 1
    with pytest.raises(RuntimeError, match="unknown option"):
        hip_ops.opt("no_such_option")


def test_no_cpu_fallback():
    from nanovllm import hip_ops
    x = torch.randn(2, 64).to(torch.bfloat16)
    with pytest.raises(RuntimeError, match="GPU tensors"):
        hip_ops.rmsnorm(x, torch.ones(64, dtype=torch.bfloat16), 1e-5)
    with pytest.raises(RuntimeError, match="not found"):
        hip_ops.load_library("/nonexistent/libglm4moe_hip.so")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "glm4_moe_nano-vllm_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in src.replace("# oracle", ""), f"{f} mentions the oracle"


def test_one_token_fusions_refuse_two_rows():
    """The fused one-token launches (combine + norm + QKV; o_proj + norm + router tail) run the FMA GEMV body, the
    stand-alone GEMM of two or more rows runs the small-M MFMA body: another fp32 summation order, so the bit-identity
    tests can only hold where BOTH sides take the same body.  r03 (gpurun_out/r03_c, commit 2982577) had the fused launch
    take T = 2 while glm_gemm_bf16 took the GEMV body for M == 1 only — `test_combine_norm_qkv_one_launch[2-...]` then
    compared two different bodies and failed; the fix (a16f249) restricted the fused launch to one row.  Pinned here."""
    from nanovllm import hip_ops
    assert hip_ops.TINYM_MAX_ROWS == 1
    y = torch.zeros(18, 64, dtype=torch.bfloat16)
    pos = torch.zeros(16, dtype=torch.int32)
    assert hip_ops.can_combine_norm_gemm(hip_ops.DeferredCombine(y, pos, 1, 8, 8, (1, 64)), 64)
    assert not hip_ops.can_combine_norm_gemm(hip_ops.DeferredCombine(y, pos, 2, 8, 16, (2, 64)), 64)
    assert hip_ops.can_gemm_router_tail(torch.zeros(1, 512, dtype=torch.bfloat16), 4096, 64)
    assert not hip_ops.can_gemm_router_tail(torch.zeros(2, 512, dtype=torch.bfloat16), 4096, 64)


def test_asm_mfma_statements_have_no_compiler_placed_hazards():
    """csrc/prefill_attn64.hip issues its MFMAs from inline asm, which the compiler's hazard recogniser cannot see: a register
    copy it places right in front of one (VALU write of SrcC) or right behind one (VALU read of the result) silently corrupts
    two of the four query tiles (seen on the chip in r04).  tools/check_mfma_hazards.py compiles the file to gfx950 ISA (no GPU
    needed) and scans for both patterns."""
    import shutil
    import subprocess
    import sys
    if not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_mfma_hazards.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 findings" in r.stdout, r.stdout[-500:]
    # and the checker does see both patterns (the sequences the failing r04 build contained)
    bad = (
        "\tv_mov_b64_e32 v[46:47], v[16:17]\n"
        "\tv_mov_b64_e32 v[44:45], v[14:15]\n"
        "\tv_mfma_f32_16x16x32_bf16 v[44:47], v[2:5], v[48:51], v[44:47]\n"
        "\tv_mov_b64_e32 v[14:15], v[44:45]\n"
        "\ts_endpgm\n")
    f = tmp_isa = os.path.join(root, "gpurun_out", "_hazard_selftest.s")
    os.makedirs(os.path.dirname(f), exist_ok=True)
    with open(f, "w") as fh:
        fh.write(bad)
    try:
        r2 = subprocess.run([sys.executable, os.path.join(root, "tools", "check_mfma_hazards.py"), tmp_isa], capture_output=True, text=True)
    finally:
        os.remove(f)
    assert r2.returncode == 1 and "(a) VALU write" in r2.stdout and "(b) v_mov_b64_e32" in r2.stdout, r2.stdout
