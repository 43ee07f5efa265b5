"""Two PROCESSES on one GPU: map each other's receive buffers over CUDA-IPC and run the one-shot all-reduce kernel
(nanovllm/distributed/oneshot_allreduce.py) — eagerly, and captured in a HIP graph that holds an ODD number of calls
(3) and is replayed 4 times with fresh inputs: the epoch lives in device memory, so every replay works on new epochs and
the two receive slots are reused out of phase.  Checks the IPC wiring and the cross-process flag protocol bit for bit
against a torch sum; xGMI itself needs two GPUs.      python tools/oneshot_allreduce_2proc.py"""
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

        def inputs(seed):
            g = torch.Generator().manual_seed(seed)
            xs = [torch.randn(32, 4096, generator=g).to(torch.bfloat16) for _ in range(world)]
            return xs, sum(x.float() for x in xs).to(torch.bfloat16)

        for rep in range(5):  # eager, odd count
            xs, exp = inputs(100 * rep)
            mine = xs[rank].cuda()
            dist.barrier()
            ar.all_reduce(mine)
            torch.cuda.synchronize()
            assert ar.status() == 0, f"rank {rank}: gave up waiting, status {ar.status()}"
            assert torch.equal(mine.cpu(), exp), f"rank {rank} eager rep {rep}"
        # ---- HIP-graph replay: 3 all-reduces per graph (odd), 4 replays ----
        bufs = [torch.zeros(32, 4096, dtype=torch.bfloat16, device="cuda") for _ in range(3)]
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            for b in bufs:
                ar.all_reduce(b)  # warm-up on the capture stream (also consumes 3 epochs on both ranks)
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for b in bufs:
                ar.all_reduce(b)
                b.mul_(0.5)       # unrelated work between the calls
        for rep in range(4):
            exps = []
            for i, b in enumerate(bufs):
                xs, exp = inputs(1000 + 10 * rep + i)
                b.copy_(xs[rank])
                exps.append((exp.float() * 0.5).to(torch.bfloat16))
            torch.cuda.synchronize()
            dist.barrier()
            g.replay()
            torch.cuda.synchronize()
            assert ar.status() == 0, f"rank {rank}: replay {rep} gave up waiting, status {ar.status()}"
            for i, b in enumerate(bufs):
                assert torch.equal(b.cpu(), exps[i]), f"rank {rank} replay {rep} call {i}"
        assert ar.epoch() == 5 + 3 + 4 * 3, f"epoch {ar.epoch()}"
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
