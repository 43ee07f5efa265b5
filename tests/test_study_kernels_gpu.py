"""The study library's kernels keep their parity tests (tests/study_kernels.py) — behind the study library: this test runs
them in a child process with GLM4MOE_HIP_LIB pointing at lib/libglm4moe_hip_study.so, and is SKIPPED when that library has
not been built (build() / the driver build the product library only; `python glm4_moe_nano-vllm_amd/build.py --study`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
STUDY_LIB = os.path.join(ROOT, "glm4_moe_nano-vllm_amd", "lib", "libglm4moe_hip_study.so")


@pytest.mark.timeout(900)
def test_study_kernels_in_the_study_library():
    if not os.path.exists(STUDY_LIB):
        pytest.skip("study library not built (build.py --study)")
    env = dict(os.environ, GLM4MOE_HIP_LIB=STUDY_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "study_kernels.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=840)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
