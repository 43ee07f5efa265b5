"""Build libglm4moe_hip.so (gfx950) in-tree with hipcc.

    python glm4_moe_nano-vllm_amd/build.py [--force]

One translation unit per kernel family under csrc/; objects are cached under build/ and
re-compiled when the source (or a header) is newer.  The shared object lands in lib/ so that it
travels with the repository snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libglm4moe_hip.so")
ARCH = "gfx950"
SOURCES = ["api", "elementwise", "moe_misc", "paged_attn", "prefill_attn", "gemm_smallm", "gemm_tinym", "gemm_largem",
           "decode_engine", "oneshot_allreduce"]
# kernels that were measured against the shipped ones and lost: only in the study library (`--study`), with their tests
STUDY_SOURCES = ["prefill_attn64", "prefill_attn_persist"]
STUDY_TAG = "_study"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}",
         "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newest_header() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "glm4moe_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(name: str, force: bool, extra=(), tag: str = "") -> str:
    src = os.path.join(CSRC, name + ".hip")
    obj = os.path.join(OBJ, name + tag + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= _newest_header()):
        return obj
    cmd = [_hipcc(), *FLAGS, *extra, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}.hip:\n{r.stdout}\n{r.stderr}")
    return obj


def build_variant(tag: str, defines: list[str], sources=None) -> str:
    """Timing-only variant library lib/libglm4moe_hip<tag>.so with extra -D flags (ablations)."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    extra = [f"-D{d}" for d in defines]
    sources = sources or (SOURCES + STUDY_SOURCES if any(d.startswith("GLM_STUDY_KERNELS=1") for d in defines) else SOURCES)
    with ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        objs = list(ex.map(lambda n: _compile(n, True, extra, tag), sources))
    out = os.path.join(LIBDIR, f"libglm4moe_hip{tag}.so")
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", out, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return out


def build_lib(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda n: _compile(n, force), SOURCES))
    if (force or not os.path.exists(LIB)
            or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[build] linked {LIB}")
    elif verbose:
        print(f"[build] up to date: {LIB}")
    return LIB


def build_study() -> str:
    """lib/libglm4moe_hip_study.so: the product sources + the measured-and-lost kernels (options fa_mfma32, fa_rows64)."""
    return build_variant(STUDY_TAG, ["GLM_STUDY_KERNELS=1"])


if __name__ == "__main__":
    if "--study" in sys.argv:
        print(build_study())
    elif "--variant" in sys.argv:  # python build.py --variant _abl1 GLM_ABLATE=1
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build_lib(force="--force" in sys.argv)
