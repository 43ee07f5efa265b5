"""What a kernel boundary costs inside the decode graph: a HIP graph of N dependent one-workgroup kernels (and the same
chain on a plain stream), time per node.  Together with the straight-line fit of the weight streams (profiles/r02/sweeps.md:
~7 us per launch) this separates the launch gap from the ramp and drain of a streaming kernel.

    python tools/launch_gap.py [--nodes 400]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
import torch

from nanovllm import hip_ops as ops


def timed(fn, reps=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=400)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "launch_gap.json"))
    args = ap.parse_args()
    ops.load_library()
    N = args.nodes
    x = torch.randn(1, 256, device="cuda").to(torch.bfloat16)
    w = torch.ones(256, device="cuda", dtype=torch.bfloat16)
    res = {}
    for name, T in (("one_workgroup", 1), ("256_workgroups", 256)):
        xx = x.expand(T, 256).contiguous()

        def chain():
            y = xx
            for _ in range(N):
                y = ops.rmsnorm(y, w, 1e-5)  # each node reads the previous node's output
            return y

        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            chain()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                chain()
            t_graph = timed(g.replay) / N
            t_stream = timed(chain, reps=5) / N
        res[name] = {"us_per_node_graph": round(t_graph, 2), "us_per_node_stream": round(t_stream, 2)}
        print(f"{name:16s}: {t_graph:.2f} us per node in a graph, {t_stream:.2f} us per node launched on a stream "
              f"({N} dependent nodes, tiny RMSNorm)")
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
