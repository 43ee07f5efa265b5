"""Decode step time vs the number of o_proj K slices (the fused norm+router reads one fp32 slab per slice)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); "
        "from nanovllm.layers.linear import RowParallelLinear; RowParallelLinear.partial_slices = int(sys.argv[1]); "
        "sys.argv = ['bench.py', '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--kernel-iters', '3']; "
        "import bench; bench.main()") % (ROOT, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
for ns in (8, 4, 6, 12):
    r = subprocess.run([sys.executable, "-c", code, str(ns)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    print(f"slices {ns:2d}: {d['value']:.1f} tokens/s, {d['ms_per_step']:.3f} ms/step")
