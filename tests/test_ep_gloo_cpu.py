"""Expert parallelism (BASELINE config 4 / SURVEY N3): dispatch -> experts -> combine over a REAL all-to-all between two
processes (gloo, CPU).  The HIP kernels cannot run here, so each rank's compute is the oracle's arithmetic behind the
hip_ops interface (tests/ep_cpu_ops.py); what is verified is what has to be right by construction for the multi-GPU run:
the pair bucketing, the split sizes of both all-to-all-v, the receive-side regrouping by expert, the un-permutation, the
ascending-expert bf16 combine and the final all-gather — EP = 2 must reproduce the TP = 1 oracle."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case(T, E, k, H, I, seed):
    g = torch.Generator().manual_seed(seed)
    bf = torch.bfloat16
    x = torch.randn(T, H, generator=g).to(bf)
    w13 = (torch.randn(E + 1, 2 * I, H, generator=g) * H ** -0.5).to(bf)
    w2 = (torch.randn(E + 1, H, I, generator=g) * I ** -0.5).to(bf)
    gate_w = (torch.randn(E, H, generator=g) * H ** -0.5).to(bf)
    bias = (torch.rand(E, generator=g) * 0.2 - 0.1).float()
    return x, w13, w2, gate_w, bias


def _worker(rank, world, port, q):
    try:
        for p in (ROOT, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        import ep_cpu_ops
        from oracle import glm4_moe_oracle as O
        from nanovllm.distributed import expert_parallel as EP
        cases = ((9, 4, 2, 64, 48, 0), (1, 8, 2, 32, 16, 1), (16, 8, 8, 64, 32, 2), (5, 2, 1, 32, 16, 3))
        # both dispatch forms: fixed-capacity slots (the default) and variable splits
        for fixed, (T, E, k, H, I, seed) in [(f, c) for f in ("1", "0") for c in cases]:
            os.environ["GLM_EP_FIXED"] = fixed
            x, w13, w2, gate_w, bias = _case(T, E, k, H, I, seed)
            idx, w = O.router(x, gate_w, bias, k, 1.0)
            idx = torch.sort(idx, dim=1)[0]  # the product's router returns ascending ids; weights re-gathered to match
            _, scores, _ = O.router_scores(x, gate_w, bias)
            sw = scores.gather(1, idx)
            w = sw / (sw.sum(dim=-1, keepdim=True) + 1e-20)
            El = E // world
            shared = lambda hh: O.mlp(hh, w13[E], w2[E])
            ex = EP.EPExperts(ep_cpu_ops, world, rank, E, k, w13[rank * El:(rank + 1) * El], w2[rank * El:(rank + 1) * El],
                              shared)
            out = EP.run(ex.forward(x, idx, w))
            ref = O.moe(x, None, None, k, 1.0, w13[:E], w2[:E], w13[E], w2[E], topk=(idx, w))
            assert out.shape == ref.shape
            # same rows, same per-row arithmetic, same combine order: at most matmul-blocking rounding apart
            err = (out.float() - ref.float()).abs().max()
            assert float(err) <= 2 ** -6 * float(ref.float().abs().max()), f"case {(T, E, k)}: max err {float(err)}"
            # every rank holds the full result
            gathered = [torch.empty_like(out) for _ in range(world)]
            dist.all_gather(gathered, out)
            assert all(torch.equal(gathered[0], t) for t in gathered)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, "FAIL: " + traceback.format_exc()))


@pytest.mark.timeout(180)
def test_ep2_dispatch_combine_over_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(30)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_ep_emulated_ranks_match_oracle():
    """The in-process driver (run_emulated) with 4 logical ranks: same collectives done by copies."""
    sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
    import ep_cpu_ops
    from oracle import glm4_moe_oracle as O
    from nanovllm.distributed import expert_parallel as EP
    T, E, k, H, I = 11, 8, 3, 64, 32
    x, w13, w2, gate_w, bias = _case(T, E, k, H, I, 7)
    idx, _ = O.router(x, gate_w, bias, k, 1.0)
    idx = torch.sort(idx, dim=1)[0]
    _, scores, _ = O.router_scores(x, gate_w, bias)
    sw = scores.gather(1, idx)
    w = sw / (sw.sum(dim=-1, keepdim=True) + 1e-20)
    ep = 4
    El = E // ep
    shared = lambda hh: O.mlp(hh, w13[E], w2[E])
    ref = O.moe(x, None, None, k, 1.0, w13[:E], w2[:E], w13[E], w2[E], topk=(idx, w))
    both = []
    for form in ("forward_fixed", "forward_varlen"):
        gens = [getattr(EP.EPExperts(ep_cpu_ops, ep, r, E, k, w13[r * El:(r + 1) * El], w2[r * El:(r + 1) * El], shared), form)(x, idx, w)
                for r in range(ep)]
        outs = EP.run_emulated(gens)
        for o in outs:
            assert torch.equal(o, outs[0])
        err = (outs[0].float() - ref.float()).abs().max()
        assert float(err) <= 2 ** -6 * float(ref.float().abs().max()), form
        both.append(outs[0])
    assert torch.equal(both[0], both[1]), "fixed-capacity and variable-split dispatch must give the same rows"


def test_ep_plan_capacity_and_overflow():
    """The stand-in plan (same layout as glm_ep_plan): the worst-case capacity never overflows; a smaller one reports it
    and drops exactly the rows beyond the slot on both sides."""
    import ep_cpu_ops
    g = torch.Generator().manual_seed(5)
    T, E, k, ep = 12, 8, 4, 4
    idx = torch.stack([torch.randperm(E, generator=g)[:k].sort()[0] for _ in range(T)])
    w = torch.rand(T, k, generator=g).to(torch.bfloat16)
    per, El = 3, 2
    cap = per * min(k, El)
    for r in range(ep):
        sg, sc, ps, ro, rg, ovf = ep_cpu_ops.ep_plan(idx, w, E, ep, r, cap)
        assert int(ovf) == 0 and int((ps >= 0).sum()) == per * k
        assert int(ro[-1]) == int((idx // El == r).sum())
    sg, sc, ps, ro, rg, ovf = ep_cpu_ops.ep_plan(idx, w, E, ep, 0, 1)
    assert int(ovf) > 1
