"""Two PROCESSES on one GPU: map each other's comm buffers over CUDA-IPC and run the one-shot all-reduce kernel
(nanovllm/distributed/oneshot_allreduce.py).  Checks the IPC wiring and the cross-process flag protocol; xGMI itself needs
two GPUs.  python tools/oneshot_allreduce_2proc.py"""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, q):
    try:
        sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from nanovllm.distributed.oneshot_allreduce import OneShotAllReduce
        n = 32 * 4096
        ar = OneShotAllReduce.for_group(n, torch.device("cuda", 0))
        for rep in range(8):
            g = torch.Generator().manual_seed(100 * rep)
            xs = [torch.randn(32, 4096, generator=g).to(torch.bfloat16) for _ in range(world)]
            exp = sum(x.float() for x in xs).to(torch.bfloat16)
            mine = xs[rank].cuda()
            dist.barrier()
            ar.all_reduce(mine)
            torch.cuda.synchronize()
            assert ar.status() == 0, f"rank {rank}: gave up waiting, status {ar.status()}"
            assert torch.equal(mine.cpu(), exp), f"rank {rank} rep {rep}"
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:
        import traceback
        q.put((rank, "FAIL " + traceback.format_exc()))


if __name__ == "__main__":
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, 29733, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=240) for _ in ps]
    for p in ps:
        p.join(30)
    print(res)
    sys.exit(0 if all(m == "ok" for _, m in res) else 1)
