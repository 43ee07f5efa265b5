import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PKG = os.path.join(ROOT, "glm4_moe_nano-vllm_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "timeout: per-test timeout (pytest-timeout)")
    # the HIP library is built in-tree (git-ignored); build it when a fresh checkout lacks it —
    # hipcc cross-compiles gfx950 without a GPU.  There is still no CPU fallback for the ops.
    lib = os.path.join(PKG, "lib", "libglm4moe_hip.so")
    if not os.path.exists(lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location("glm4moe_build", os.path.join(PKG, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_lib(force=False, verbose=False)


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped, never silently passed, when no GPU is visible
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
