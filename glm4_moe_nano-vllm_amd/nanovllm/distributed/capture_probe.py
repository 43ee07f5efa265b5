"""Pre-flight for HIP-graph capture of RCCL collectives (TP > 1).

A decode step with its all-reduces is worth capturing (an eager TP step costs ~0.19 ms of host time per layer), but a
capture that fails leaves the process unusable and the other ranks blocked in a collective.  So the question "can this
node capture and replay all-reduces issued through torch.distributed?" is asked in a THROWAWAY child process per rank,
on its own rendezvous port, BEFORE the rank brings up its own communicator; the ranks then agree on the answer
(all-reduce MIN) and fall back to eager launches together when any child failed.

    python -m nanovllm.distributed.capture_probe      (child: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env)
"""
from __future__ import annotations

import os
import subprocess
import sys

PORT_OFFSET = 17  # the children rendezvous on MASTER_PORT + 17


def child_main() -> int:
    """Exit code 0 = two captured all-reduces replayed with the right sums."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local_rank = 0 if os.environ.get("GLM_BENCH_SINGLE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", device_id=dev)
    x = torch.full((32, 4096), float(rank + 1), dtype=torch.bfloat16, device=dev)
    dist.all_reduce(x)  # brings the communicator up outside the capture
    torch.cuda.synchronize()
    y = torch.full((32, 4096), float(rank + 1), dtype=torch.bfloat16, device=dev)
    g = torch.cuda.CUDAGraph()
    # the expert-parallel dispatch adds an all-to-all with equal splits and an all-gather (distributed/expert_parallel.py):
    # probed only when the run will use them (GLM_PROBE_EP=1), so a plain TP run never loses its graph to them
    ep = os.environ.get("GLM_PROBE_EP") == "1"
    z = torch.arange(world * 8, dtype=torch.float32, device=dev).view(world, 8) + 100.0 * rank
    zo, zg = torch.empty_like(z), torch.empty(world * 8, dtype=torch.float32, device=dev)
    if ep:
        dist.all_to_all_single(zo, z)
        dist.all_gather_into_tensor(zg, z[0].contiguous())
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        a = y * 2
        dist.all_reduce(a)
        b = a + 1
        dist.all_reduce(b)
        if ep:
            dist.all_to_all_single(zo, z)
            dist.all_gather_into_tensor(zg, z[0].contiguous())
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    s1 = 2.0 * sum(r + 1 for r in range(world))
    ok = bool((a == s1).all()) and bool((b == world * (s1 + 1)).all())
    exp_a2a = torch.stack([torch.arange(8, dtype=torch.float32, device=dev) + 8 * rank + 100.0 * s for s in range(world)])
    exp_ag = torch.cat([torch.arange(8, dtype=torch.float32, device=dev) + 100.0 * s for s in range(world)])
    if ep:
        ok = ok and bool((zo == exp_a2a).all()) and bool((zg == exp_ag).all())
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


def probe(env: dict | None = None, timeout_s: int = 180, argv: list[str] | None = None) -> tuple[bool, str | None]:
    """Parent side: run the child with this rank's environment on the next rendezvous port.  Starts a child process —
    never replaces the running one — so it is safe after this process has initialised its GPU."""
    env = dict(os.environ if env is None else env)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + PORT_OFFSET)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    pkg_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env["PYTHONPATH"] = pkg_root + os.pathsep + env.get("PYTHONPATH", "")
    cmd = argv or [sys.executable, "-m", "nanovllm.distributed.capture_probe"]
    try:
        r = subprocess.run(cmd, env=env, timeout=timeout_s, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    except subprocess.TimeoutExpired:
        return False, "capture probe timed out"
    if r.returncode != 0:
        tail = r.stderr.decode(errors="replace").strip().splitlines()[-1:] or [""]
        return False, f"capture probe exit {r.returncode}: {tail[0][:160]}"
    return True, None


def agree(ok: bool, device) -> bool:
    """Every rank takes the same decision: captured only if every rank's child succeeded."""
    import torch
    import torch.distributed as dist
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


if __name__ == "__main__":
    sys.exit(child_main())
