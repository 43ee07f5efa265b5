#!/usr/bin/env python3
"""bench.py — decode throughput of the MI355X-native GLM-4-MoE layer path.

Workload (BASELINE.json configs[2], the one the metric is quoted on): GLM-4-MoE "Air-64E"
(H=4096, 96 q / 8 kv heads x 128, 64 routed experts top-8 + 1 shared, I_moe=1408, dense layer 0
with I=10944), random-init bf16 weights, synthetic KV cache, greedy decode of B=32 sequences at
context 4096, TP = number of GPUs.  One "step" = one decode step of the whole model (embed →
L decoder layers → final norm → lm_head → argmax) for the batch = B tokens.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--layers L] [--batch B] [--ctx C]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (grouped gate|up expert
GEMM, HBM-bound at this shape) with its duration measured live by events on the launch stream;
`kernels` lists the same for the other hot kernels; `cpu_baseline` times the CPU oracle (the
reference's eager algorithm, incl. its per-expert Python loop) for one decoder layer on the
host cores of the same box.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "glm4_moe_nano-vllm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def air64e_config(layers: int, max_pos: int):
    return types.SimpleNamespace(
        hidden_size=4096, num_attention_heads=96, num_key_value_heads=8, head_dim=128,
        attention_bias=True, use_qk_norm=False, partial_rotary_factor=0.5, rope_theta=1000000.0,
        rope_scaling=None, max_position_embeddings=max_pos, rms_norm_eps=1e-5,
        n_routed_experts=64, num_experts_per_tok=8, n_group=1, topk_group=1,
        routed_scaling_factor=1.0, moe_intermediate_size=1408, intermediate_size=10944,
        first_k_dense_replace=1, hidden_act="silu", vocab_size=151552,
        num_hidden_layers=layers, num_layers_to_load=layers, tie_word_embeddings=False,
        torch_dtype=torch.bfloat16)


def init_weights(model, seed: int):
    """Every tensor from its own generator seeded by (seed, name): replicated tensors (router, norms, embedding) agree on
    every rank even when the ranks own different parameter sets (expert parallelism)."""
    import zlib
    g = torch.Generator(device="cuda")
    for name, p in model.named_parameters():
        g.manual_seed(seed + zlib.crc32(name.encode()))
        if "layernorm" in name or name.endswith("norm.weight"):
            p.data.normal_(0.0, 0.02, generator=g).add_(1.0)
        else:
            p.data.normal_(0.0, 0.02, generator=g)
    for name, b in model.named_buffers():
        if name.endswith("e_score_correction_bias"):
            g.manual_seed(seed + zlib.crc32(name.encode()))
            b.uniform_(-0.1, 0.1, generator=g)


def probe_capture_child() -> int:
    """Child process of the N>1 pre-flight (nanovllm/distributed/capture_probe.py: can this node capture RCCL
    all-reduces in a HIP graph and replay them?).  Exit code 0 = yes."""
    from nanovllm.distributed import capture_probe
    return capture_probe.child_main()


def probe_capture(world: int, timeout_s: int = 180) -> tuple[bool, str | None]:
    """Parent side: run the probe child with this rank's environment on the next rendezvous port."""
    from nanovllm.distributed import capture_probe
    return capture_probe.probe(timeout_s=timeout_s, argv=[sys.executable, os.path.abspath(__file__), "--probe-capture"])


def spawn_ranks(n: int) -> int:
    """Launcher side of `python bench.py --gpus N` without a distributed environment: N ranks of this same file as
    children of `python -m torch.distributed.run` (one rank per GPU, rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=46)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-c2", action="store_true", help="skip the second measured entry (BASELINE config 2: bs=1 ctx=1024)")
    ap.add_argument("--kernel-iters", type=int, default=20)
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `extra` measurements after the timed region (fp8 experts, prefill step / attention / dense GEMMs)")
    ap.add_argument("--emulate-tp", type=int, default=1,
                    help="kernel-shape study on ONE GPU: run the per-rank shapes of TP=N (heads, kv heads, expert/MLP "
                         "intermediate size and vocabulary divided by N) without any collective; not a benchmark result")
    ap.add_argument("--expert-parallel", action="store_true",
                    help="BASELINE config 4 flavour: the experts are partitioned over the ranks (E/N each, full intermediate "
                         "size) and token rows travel by an all-to-all of fixed-capacity slots planned on the device "
                         "(graph-capturable); attention stays tensor-parallel.  The headline run is the default (TP only)")
    ap.add_argument("--probe-capture", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--graph-body-only", action="store_true",
                    help="debug: at 1 GPU, capture only the layer stack and run the lm_head eagerly (the TP>1 arrangement)")
    ap.add_argument("--fp8-experts", action="store_true",
                    help="stream the experts as OCP fp8 e4m3 (weight-only, N4 / BASELINE config 5 flavour); "
                         "the default bf16 run is the one comparable with the reference")
    args = ap.parse_args()
    if args.probe_capture:
        sys.exit(probe_capture_child())

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched a GPU (importing
        # torch does not) and never will: it starts the N ranks as fresh children through torch.distributed.run, lets
        # rank 0's JSON line through on the inherited stdout and exits with the children's code.
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # rehearsal knobs for a one-GPU box (never set by the driver): all ranks on cuda:0 over gloo exercises this
    # file's N>1 control flow (sharded build, collectives' order, rank-0-only sections) without RCCL
    backend = os.environ.get("GLM_BENCH_BACKEND", "nccl")
    if os.environ.get("GLM_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    # N>1 pre-flight, before this process touches its GPU: may the layer stack be captured with its collectives?
    tp_graph_ok, tp_graph_note = True, None
    if args.expert_parallel:
        os.environ["GLM_PROBE_EP"] = "1"  # the capture pre-flight then also captures an all-to-all and an all-gather
    if world > 1 and not args.no_graph:
        if backend != "nccl" and os.environ.get("GLM_BENCH_FORCE_PROBE") != "1":
            tp_graph_ok, tp_graph_note = False, f"{backend} collectives cannot be captured"
        else:
            tp_graph_ok, tp_graph_note = probe_capture(world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        flag = torch.tensor([1 if tp_graph_ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # every rank takes the same decision
        if not int(flag.item()) and tp_graph_ok:
            tp_graph_ok, tp_graph_note = False, "capture probe failed on another rank"

    from nanovllm import hip_ops
    from nanovllm.models.glm4_moe.model import Glm4MoeForCausalLM
    from nanovllm.utils.context import set_context
    hip_ops.load_library()

    B, ctx, L, bs = args.batch, args.ctx, args.layers, 256
    cfg = air64e_config(L, max(ctx, 4096))
    if args.emulate_tp > 1:
        n = args.emulate_tp
        cfg.num_attention_heads //= n
        cfg.num_key_value_heads = max(1, cfg.num_key_value_heads // n)
        cfg.moe_intermediate_size //= n
        cfg.intermediate_size //= n
        cfg.vocab_size //= n
    if args.fp8_experts:
        cfg.expert_weight_dtype = "fp8_e4m3"
    if args.expert_parallel and world > 1:
        cfg.expert_parallel_size = world  # fixed-capacity dispatch (no host read): captured like the TP step
    torch.set_default_dtype(torch.bfloat16)
    t0 = time.time()
    with torch.device(dev):
        model = Glm4MoeForCausalLM(cfg)
    init_weights(model, 1234)  # same seed on every rank: replicated tensors (router!) agree
    torch.cuda.synchronize()
    t_build = time.time() - t0

    # ---- paged KV cache (model_runner.py:108-125 layout), pre-filled N(0,1) ----
    nkv_local = max(1, cfg.num_key_value_heads // world)
    blocks_per_seq = (ctx + bs - 1) // bs
    nblk = B * blocks_per_seq + 8
    kv = torch.empty(2, L, nblk, bs, nkv_local, cfg.head_dim, device=dev)
    for l in range(L):
        kv[:, l].normal_(0.0, 1.0)
    lid = 0
    for m in model.modules():
        if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
            m.k_cache, m.v_cache = kv[0, lid], kv[1, lid]
            lid += 1
    g = torch.Generator().manual_seed(7)
    perm = torch.randperm(nblk, generator=g)[: B * blocks_per_seq].view(B, blocks_per_seq)
    block_tables = perm.to(torch.int32).to(dev)
    context_lens = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)
    slot_mapping = (block_tables[:, (ctx - 1) // bs].to(torch.int64) * bs + (ctx - 1) % bs).to(torch.int32)
    input_ids = torch.randint(0, cfg.vocab_size, (B,), generator=g).to(dev)
    set_context(False, slot_mapping=slot_mapping, context_lens=context_lens, block_tables=block_tables)
    out_tokens = torch.zeros(B, dtype=torch.int64, device=dev)

    hidden = {}

    @torch.inference_mode()
    def body():
        hidden["h"] = model(input_ids, positions)

    @torch.inference_mode()
    def head():
        logits = model.compute_logits(hidden["h"])
        if logits is not None:
            out_tokens.copy_(torch.argmax(logits, dim=-1))

    def step():
        body()
        head()

    # ---- warm-up (also sets kernel attributes, sizes workspaces, brings up the RCCL communicator), then
    # HIP-graph capture.  TP=1: the whole step is one graph.  TP>1: the layer stack with its RCCL all-reduces is
    # captured (rehearsed on one GPU by tools/tp_graph_rehearsal.py; an eager TP step needs ~0.19 ms of host time
    # per layer, more than its GPU time at TP>=4); the lm_head gather stays eager.  Ranks agree on the outcome,
    # and any failure falls back to eager on every rank.
    from nanovllm.distributed.parallel_state import get_tp_group
    get_tp_group().reset_counters()
    step()
    torch.cuda.synchronize()
    collective = collective_record(get_tp_group(), world, dev) if world > 1 else None
    use_graph = not args.no_graph and (world == 1 or tp_graph_ok)
    whole = world == 1 and not args.graph_body_only
    graph_note = tp_graph_note
    if use_graph:
        ok = 1
        graph = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(graph):
                if whole:
                    step()
                else:
                    body()
        except Exception as e:  # noqa: BLE001 - report and fall back
            ok, graph_note = 0, repr(e)[:200]
            # torch.cuda.graph.__exit__ raised from capture_end() before restoring the stream: leave the (ended,
            # invalidated) capture stream, otherwise every later call fails with "stream is capturing"
            torch.cuda.set_stream(torch.cuda.default_stream(dev))
            graph = None
            try:
                torch.cuda.synchronize()
            except Exception:  # noqa: BLE001
                pass
        if world > 1:
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = int(flag.item())
        use_graph = bool(ok)
    if use_graph and whole:
        run = graph.replay
    elif use_graph:
        def run():
            graph.replay()
            head()
    else:
        run = step
    for _ in range(args.warmup):
        run()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    tok_s = B * args.steps / dt

    # ---- live per-kernel durations (events on the launch stream), algorithmic bytes ----
    kernels = {}
    if rank == 0:
        kernels = measure_kernels(model, cfg, hip_ops, B, ctx, world, args.kernel_iters, dev)

    # ---- BASELINE config 2 (greedy bs=1, ctx=1024) on the same model, as a second measured entry ----
    extra = {}
    if world == 1 and not args.no_c2 and (B, ctx) != (1, 1024) and args.emulate_tp == 1:
        extra["c2_bs1_ctx1024"] = measure_c2(model, cfg, kv, dev, L, bs, args.fp8_experts)
        if not args.fp8_experts and hip_ops.opt("t1_engine") == 0:
            # r05: the same step with the one-launch MLP half (glm_decode_mlp_engine, opt-in): measured beside the default
            # five-launch layer so that the driver line carries the comparison
            try:
                with hip_ops.option("t1_engine", 1):
                    eng = measure_c2(model, cfg, kv, dev, L, bs, False)
                hip_ops.check_decode_engine(dev)
                extra["c2_bs1_ctx1024"]["one_launch_mlp_half"] = {
                    "tokens_per_s": eng["tokens_per_s"], "ms_per_layer": eng["ms_per_layer"], "layer_hbm_frac": eng["layer_hbm_frac"],
                    "launches_per_moe_layer": 3, "default_launches_per_moe_layer": 5,
                    "note": "option t1_engine = 1: o_proj .. down projection as ONE persistent launch, bit-identical; default off"}
            except Exception as e:  # noqa: BLE001 - a side leg never takes the headline line down
                extra["c2_bs1_ctx1024"]["one_launch_mlp_half"] = {"error": repr(e)[:200]}
        set_context(False, slot_mapping=slot_mapping, context_lens=context_lens, block_tables=block_tables)

    # ---- the other numbers README.md / DESIGN.md quote, on the driver line (rank 0, TP=1, after the timed region):
    # prefill attention, dense prefill projections, a whole prefill step over 3 layers ----
    if rank == 0 and world == 1 and not args.no_extras and args.emulate_tp == 1 and not args.fp8_experts:
        extra.update(measure_extras(model, cfg, hip_ops, dev, L))
        set_context(False, slot_mapping=slot_mapping, context_lens=context_lens, block_tables=block_tables)

    cpu_base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_base = cpu_baseline(model, cfg, kv, block_tables, context_lens, positions, slot_mapping, B, L)
        try:  # BASELINE configs[0] beside it ("for config 1 and config 3")
            cpu_base["c1"] = cpu_baseline_c1(hip_ops, dev, cpu_base["cores"])
        except Exception as e:  # noqa: BLE001 - a side leg never takes the headline line down
            cpu_base["c1"] = {"error": repr(e)[:200]}
        set_context(False, slot_mapping=slot_mapping, context_lens=context_lens, block_tables=block_tables)  # the C1 leg reset it

    # ---- fp8 (e4m3, weight-only) experts on the same model, re-quantised in place: LAST, because the bf16 expert
    # stacks are released by the quantisation ----
    if rank == 0 and world == 1 and not args.no_extras and args.emulate_tp == 1 and not args.fp8_experts:
        try:
            extra["fp8_experts"] = measure_fp8(model, cfg, hip_ops, step, B, ctx, L, args, dev)
        except Exception as e:  # noqa: BLE001 - an extra never takes the headline line down
            import traceback
            traceback.print_exc(file=sys.stderr)
            extra["fp8_experts"] = {"error": repr(e)[:200]}

    if rank == 0:
        attach_pmc_traffic(kernels, {"world": world, "emulate_tp": args.emulate_tp,
                                     "expert_parallel": bool(args.expert_parallel and world > 1)})
        dom = kernels.get("grouped_gemm_gateup_silu")
        line = {
            "metric": ("decode tokens/sec/GPU GLM-4-MoE bs=32 seq=4096; grouped-GEMM MFMA util %" if args.emulate_tp == 1 else
                       f"SHAPE STUDY ONLY: one rank's kernels at TP={args.emulate_tp} shapes, no collectives"),
            "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16 (experts: fp8 e4m3 weight-only)" if args.fp8_experts else "bf16", "data": "synthetic",
            "config": {"workload": f"GLM-4-MoE Air-64E decode bs={B} ctx={ctx} L={L} (layer0 dense), "
                                   f"random-init bf16, paged KV block 256, greedy",
                       "global_batch": B, "seq_len": ctx, "layers": L,
                       "parallelism": f"tp{world}" + (f"+ep{world}" if getattr(cfg, "expert_parallel_size", 1) > 1 else ""),
                       "hip_graph": bool(use_graph), "hip_graph_note": graph_note,
                       "ms_per_layer": round(ms_per_step / L, 4), "model_build_s": round(t_build, 1),
                       # every launch-path switch of both sides, as read ONCE when the library was loaded
                       "options": hip_ops.options()},
            "roofline": dom,
            "kernels": kernels,
            "extra": extra,
            "cpu_baseline": cpu_base,
        }
        if collective is not None:
            if getattr(cfg, "expert_parallel_size", 1) > 1:
                from nanovllm.distributed.expert_parallel import EPExperts
                collective["expert_parallel"] = {"capacity_factor": EPExperts.capacity_factor,
                                                 "overflowed_slots_worst_rows": EPExperts.overflowed(reset=False),
                                                 "note": "bench.py runs the worst-case capacity (no factor): an overflow is impossible; "
                                                         "ModelRunner applies the factor from a mean slot load of 16 rows on and counts redone steps"}
            line["collective"] = collective
        print(json.dumps(line))
    if world > 1:
        dist.barrier()  # ranks > 0 wait here while rank 0 finishes its per-kernel measurements
        dist.destroy_process_group()


def _kernel_source_digest() -> str:
    """sha256 over the kernel sources and this file: what a PMC summary must have been taken with to describe this run."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "glm4_moe_nano-vllm_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def _git(*args) -> str | None:
    import subprocess
    try:
        r = subprocess.run(["git", *args], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=10)
        return r.stdout.decode().strip() or None if r.returncode == 0 else None
    except Exception:  # noqa: BLE001 - no git on the box is not an error
        return None


def measure_c2(model, cfg, kv, dev, L, bs, fp8, steps=30, warmup=5):
    """BASELINE config 2 — random-init bf16, TP=1, greedy decode of ONE sequence at context 1024 — on the model the
    headline run built: whole step in one HIP graph, tokens/s, and the layer's algorithmic HBM bytes over its time."""
    import torch
    from nanovllm.utils.context import set_context
    B, ctx = 1, 1024
    per = (ctx + bs - 1) // bs
    bt = torch.arange(3, 3 + per, dtype=torch.int32, device=dev).view(1, per)
    cl = torch.full((B,), ctx, dtype=torch.int32, device=dev)
    pos = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)
    slots = (bt[:, (ctx - 1) // bs].to(torch.int64) * bs + (ctx - 1) % bs).to(torch.int32)
    ids = torch.randint(0, cfg.vocab_size, (B,), generator=torch.Generator().manual_seed(11)).to(dev)
    tok = torch.zeros(B, dtype=torch.int64, device=dev)
    set_context(False, slot_mapping=slots, context_lens=cl, block_tables=bt)

    @torch.inference_mode()
    def step():
        logits = model.compute_logits(model(ids, pos))
        tok.copy_(torch.argmax(logits, dim=-1))

    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(warmup):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    H, I, E, k = cfg.hidden_size, cfg.moe_intermediate_size, cfg.n_routed_experts, cfg.num_experts_per_tok
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    wb = 1 if fp8 else 2
    layer_bytes = ((nh + 2 * nkv) * hd * H * 2 + H * nh * hd * 2 + E * H * 2 + (k + 1) * 3 * I * H * wb
                   + B * ctx * nkv * hd * 2 * 2)
    return {"workload": "greedy decode bs=1 ctx=1024, same model", "tokens_per_s": round(B / dt, 2),
            "ms_per_step": round(dt * 1e3, 4), "ms_per_layer": round(dt * 1e3 / L, 4), "steps": steps, "hip_graph": True,
            "layer_algorithmic_bytes": int(layer_bytes),
            "layer_hbm_frac": round(layer_bytes / (dt / L) / 1e9 / HBM_PEAK_GBS, 4),
            "bytes_model": "QKV + O + router + (k+1) experts x 3*I*H + KV read, per MoE layer (BASELINE.md §3 at bs=1)"}


def collective_record(tp, world: int, dev) -> dict:
    """What the collectives of ONE eager step were, from the live process group — so that an N > 1 line proves the
    backend saw N ranks: every rank all-reduces a one (sum must be N) and all-gathers its own view of the world size;
    the call and byte counts are the TP coordinator's own counters over the step that has just run (all-reduce sites:
    layers/linear.py:153-154 of the reference, o_proj and the MLP / MoE down projection of every layer, plus the
    vocabulary-parallel embedding).  Collective: every rank must call it."""
    one = torch.ones(1, dtype=torch.int32, device=dev)
    dist.all_reduce(one, group=tp.process_group)
    mine = torch.tensor([dist.get_world_size(tp.process_group), dist.get_rank(tp.process_group), int(one.item())],
                        dtype=torch.int32, device=dev)
    seen = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(seen, mine, group=tp.process_group)
    views = [[int(v) for v in s.tolist()] for s in seen]
    return {"backend": dist.get_backend(tp.process_group), "world": dist.get_world_size(tp.process_group),
            "allreduce_per_step": int(tp.calls), "bytes_per_allreduce": int(tp.bytes_max),
            "allreduce_bytes_per_step": int(tp.bytes_total), "oneshot": bool(tp.calls_oneshot),
            "oneshot_calls_per_step": int(tp.calls_oneshot),
            # per rank: [world size it sees, its rank, sum of ones over the group] — all three must say N / r / N
            "rccl_world_per_rank": views,
            "ranks_agree": all(v[0] == world and v[2] == world for v in views) and sorted(v[1] for v in views) == list(range(world)),
            "measured_on_hardware_note": "filled from the live group of THIS run; with GLM_BENCH_BACKEND=gloo it is a rehearsal"}


def _pmc_parallelism(meta: dict) -> dict:
    """The parallelism a PMC summary was taken at: recorded in `_meta.parallelism` since r05; older files carry only the
    command line, from which the same three facts are read."""
    if "parallelism" in meta:
        return meta["parallelism"]
    cmd = (meta.get("command") or "").split()

    def flag(name, default):
        return int(cmd[cmd.index(name) + 1]) if name in cmd and cmd.index(name) + 1 < len(cmd) else default
    return {"world": flag("--gpus", 1), "emulate_tp": flag("--emulate-tp", 1), "expert_parallel": "--expert-parallel" in cmd}


def attach_pmc_traffic(kernels: dict, parallelism: dict | None = None) -> None:
    """`traffic` = HBM bytes per launch from the PMC counters.  The counters need their own `rocprofv3 --pmc` passes
    (FETCH_SIZE and WRITE_SIZE cannot share a pass, and a profiled run is not a timed run), so they come from the
    newest committed summary under profiles/ (tools/pmc_traffic.sh, same workload, TP=1).  The summary records the
    digest of the kernel sources it was taken with; it is attached only when that digest equals the digest of the
    sources of THIS run AND the parallelism recorded there (`_meta.parallelism`: world, emulate_tp, expert_parallel) is this
    run's — a TP = N rank launches other shapes than the TP = 1 run the counters saw —, together with the file's git blob id
    and the commit that last touched it; otherwise `traffic` stays null and `traffic_note` says why."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))
    if not files:
        return
    with open(files[-1]) as f:
        pmc = json.load(f)
    meta = pmc.get("_meta", {})
    rel = os.path.relpath(files[-1], ROOT)
    now = _kernel_source_digest()
    fresh = meta.get("kernel_source_digest") == now
    parallelism = parallelism or {"world": 1, "emulate_tp": 1, "expert_parallel": False}
    pmc_par = _pmc_parallelism(meta)
    same_shapes = all(pmc_par.get(k) == parallelism.get(k) for k in ("world", "emulate_tp", "expert_parallel"))
    src = {"file": rel, "git_blob": _git("hash-object", files[-1]), "git_commit": _git("log", "-1", "--format=%h", "--", rel),
           "kernel_source_digest": meta.get("kernel_source_digest"), "this_run_digest": now}
    for name, rec in kernels.items():
        if name in pmc and (rec.get("bound") == "hbm" or "hbm_bytes_corrected" in pmc[name]):
            if fresh and same_shapes:
                rec["traffic"] = pmc[name]["hbm_bytes_corrected"]
                rec["traffic_source"] = src
            elif fresh:  # per-rank kernel shapes differ (TP = N halves ... the intermediate size): not this run's bytes
                rec["traffic_note"] = f"{rel} was taken at {pmc_par}, this run is {parallelism}: not attached"
            else:
                rec["traffic_note"] = f"{rel} was taken with other kernel sources ({src['kernel_source_digest']} != {now})"


def measure_kernels(model, cfg, ops, B, ctx, world, iters, dev, only_experts=False):
    """Time each hot kernel alone on the current stream with events, rotating over the layers'
    weights so every launch streams cold data (each launch touches >> the 256 MiB L3)."""
    import torch
    layers = [l for l in model.model.layers if hasattr(l.mlp, "gate")]
    if not layers:
        return {}
    H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size // world
    nh, nkv, hd = cfg.num_attention_heads // world, max(1, cfg.num_key_value_heads // world), cfg.head_dim
    x = torch.randn(B, H, device=dev).to(torch.bfloat16)
    from nanovllm.utils.context import get_context
    c = get_context()

    def timeit(fn, n):
        """n back-to-back launches between two events on the launch stream (the queue stays
        full, so host launch gaps are not counted); repeated 3x, returns (mean, best) per launch."""
        fn(0)
        torch.cuda.synchronize()
        per = []
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(n):
                fn(rep * n + i)
            b.record()
            torch.cuda.synchronize()
            per.append(a.elapsed_time(b) / n * 1e3)
        return sum(per) / len(per), min(per)  # us

    res = {}
    # routing shared by the expert GEMMs (router output differs per layer; use layer i's gate)
    prep = []
    fp8 = getattr(cfg, "expert_weight_dtype", None) == "fp8_e4m3"
    for l in layers:
        idx, w = l.mlp.gate(x)
        offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
        if fp8 and l.mlp._fp8 is None:
            l.mlp.quantize_experts_fp8()
        w13, w2 = (None, None) if fp8 else l.mlp._stacked()  # fp8 experts have released their bf16 stacks
        prep.append((offs, tok, scale, pos, w13, w2))
    torch.cuda.synchronize()
    R = B * k + B
    touched = float(torch.stack([(p[0][1:E + 1] - p[0][:E] > 0).sum() for p in prep]).float().mean()) + 1
    nl = len(layers)

    def rec(name, mean_us, med_us, nbytes, note):
        res[name] = {"bound": "hbm", "achieved": round(nbytes / (mean_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(nbytes / (mean_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": None, "kernel_us_mean": round(mean_us, 2), "kernel_us_best": round(med_us, 2),
                     "algorithmic_bytes": int(nbytes), "bytes_model": note}

    acts = [None] * nl

    wb = 1 if fp8 else 2  # bytes per expert weight element

    def f_gateup(i):
        p = prep[i % nl]
        if fp8:
            q = layers[i % nl].mlp._fp8
            acts[i % nl] = ops.grouped_gemm_gateup_silu_fp8w(x, q[0], q[1], p[0], p[1], R)
        else:
            acts[i % nl] = ops.grouped_gemm_gateup_silu(x, p[4], p[0], p[1], R)
    m, md = timeit(f_gateup, iters)
    rec("grouped_gemm_gateup_silu", m, md, touched * 2 * I * H * wb,
        f"touched groups ({touched:.1f} incl. shared) x 2*I*H*{wb} B of gate|up weights")

    def f_down(i):
        p = prep[i % nl]
        a = acts[i % nl] if acts[i % nl] is not None else acts[0]
        if fp8:
            q = layers[i % nl].mlp._fp8
            ops.grouped_gemm_down_fp8w(a, q[2], q[3], p[0], p[2], E)
        else:
            ops.grouped_gemm_down(a, p[5], p[0], p[2], E)
    m, md = timeit(f_down, iters)
    rec("grouped_gemm_down", m, md, touched * H * I * wb, f"touched groups x H*I*{wb} B of down weights")
    if only_experts:
        return res

    attn_layers = [l.self_attn for l in model.model.layers]
    q = torch.randn(B, nh, hd, device=dev).to(torch.bfloat16)

    # the attention launch AS THE CAPTURED STEP RUNS IT (models/glm4_moe/attention_new.py): rotary + K/V cache store +
    # paged attention in one launch, fed by the QKV projection's K-slice partial sums at decode batch sizes (reduced +
    # biased on load) or by its bf16 rows otherwise; the input is produced once per layer outside the timed launches
    a0 = attn_layers[0]
    fused_ok = (a0.fuse_decode_attention and not a0.use_qk_norm and hd == 128 and a0.rotary_emb.rotary_dim == 64)
    if fused_ok:
        feeds = []
        for a in attn_layers:
            if a.defer_qkv_reduce and a.qkv_proj.can_defer_reduce(x):
                feeds.append((a.qkv_proj.forward_partial(x), a.qkv_proj.bias.data if a.qkv_proj.bias is not None else None))
            else:
                feeds.append((a.qkv_proj(x), None))
        pos = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)

        def f_attn_step(i):
            a = attn_layers[i % len(attn_layers)]
            qkv, bias = feeds[i % len(attn_layers)]
            ops.paged_attn_decode_qkv(qkv, nh, nkv, hd, pos, a.rotary_emb.cos_sin_cache, a.rotary_emb.rotary_dim, c.slot_mapping,
                                      a.attn.k_cache, a.attn.v_cache, c.context_lens, c.block_tables, a.attn.scale, qkv_bias=bias)
        m, md = timeit(f_attn_step, iters)
        rec("paged_attn_decode", m, md, B * ctx * nkv * hd * 2 * 2,
            "B*ctx*nkv*hd*2(K,V)*2 B; the launch the step runs: rotary + K/V store + attention from the QKV "
            + ("K-slice partial sums" if isinstance(feeds[0][0], ops.PartialSum) else "rows"))
        del feeds

    def f_attn(i):
        a = attn_layers[i % len(attn_layers)].attn
        ops.paged_attn_decode(q, a.k_cache, a.v_cache, c.context_lens, c.block_tables, a.scale)
    m, md = timeit(f_attn, iters)
    rec("paged_attn_decode" if not fused_ok else "paged_attn_decode_plain", m, md, B * ctx * nkv * hd * 2 * 2,
        "B*ctx*nkv*hd*2(K,V)*2 B; attention alone on rotated bf16 Q rows (layers/attention.py operator)")

    def f_qkv(i):  # as the decoder layer runs it: K-slice partial sums from 17 rows on (reduced by the attention's loads)
        a = attn_layers[i % len(attn_layers)]
        if a.defer_qkv_reduce and a.qkv_proj.can_defer_reduce(x):
            a.qkv_proj.forward_partial(x)
        else:
            a.qkv_proj(x)
    m, md = timeit(f_qkv, iters)
    rec("gemm_qkv", m, md, (nh + 2 * nkv) * hd * H * 2, "(nh+2nkv)*hd*H*2 B of W_qkv")

    o_in = torch.randn(B, nh * hd, device=dev).to(torch.bfloat16)

    def f_o(i):  # as the decoder layer runs it: K-slice partial sums at TP=1 (reduced inside the next add-RMSNorm)
        a = attn_layers[i % len(attn_layers)]
        if world == 1:
            a.o_proj(o_in, defer_reduce=True)
        else:  # the module would all-reduce: only rank 0 runs these measurements, so call the kernel itself
            ops.gemm_bf16(o_in, a.o_proj.weight.data, None)
    m, md = timeit(f_o, iters)
    rec("gemm_o", m, md, H * nh * hd * 2, "H*nh*hd*2 B of W_o")
    # ---- prefill shape of the same config: T = 16384 tokens (max_num_batched_tokens,
    # config.py:9) → 131072 routed pairs + 16384 shared rows; MFMA-bound ----
    if not fp8:  # the prefill-shape MFMA measurement is for the bf16 tiled kernel
        try:
            res.update(measure_prefill_gemm(layers[0], cfg, ops, world, dev))
        except torch.OutOfMemoryError:
            pass
    return res


MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 dense


FP8_MFMA_PEAK_TFLOPS = 5000.0  # MI355X_MICROARCH.md: ~5 PF dense fp8 (block-scaled MFMA, 2x the bf16 rate)


def measure_prefill_gemm(layer, cfg, ops, world, dev, T=16384, iters=10, warm=5, fp8=False, a8=False):
    """fp8: weight-only (fp8 bytes as the bf16 MFMA kernel's weight operand).  a8 (r05): OPT-IN W8A8 — activations quantised to
    e4m3 too (per token for gate|up, MX per 32 columns for down, produced by the gate|up epilogue), both GEMMs on the
    block-scaled fp8 MFMA; the activation quantisation launch is timed with the pair and reported."""
    import torch
    H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size // world
    x = torch.randn(T, H, device=dev).to(torch.bfloat16)
    idx, w = layer.mlp.gate(x)
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    R = T * k + T
    out = {}

    def timed(fn):
        # the chip needs a few launches of MFMA-bound work to settle its clock after the HBM-bound decode kernels
        # (tools/prefill_gemm_sustained.py: 999 TFLOP/s over the first five launches, 1046-1054 over the next 55):
        # warm up first, then time a sustained stretch
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    peak = MFMA_PEAK_TFLOPS
    quant_us = None
    if a8:
        peak = FP8_MFMA_PEAK_TFLOPS
        w13_q, w13_s, w2_q, w2_s = layer.mlp._fp8
        xq, xs = ops.quantize_fp8_rows(x)
        gu = lambda: ops.grouped_gemm_gateup_silu_fp8a8(xq, xs, w13_q, w13_s, offs, tok, R)
        r = gu()
        if r is None:
            raise RuntimeError("the fp8 x fp8 kernel does not cover this shape")
        act_q, act_bs = r
        dn = lambda: ops.grouped_gemm_down_fp8a8(act_q, act_bs, w2_q, w2_s, offs, scale, E)
        quant_us = timed(lambda: ops.quantize_fp8_rows(x))
    elif fp8:  # the fp8 bytes are the tiled kernel's weight operand (no expansion pass)
        w13_q, w13_s, w2_q, w2_s = layer.mlp._fp8
        gu = lambda: ops.grouped_gemm_gateup_silu_fp8w(x, w13_q, w13_s, offs, tok, R)
        act = gu()
        dn = lambda: ops.grouped_gemm_down_fp8w(act, w2_q, w2_s, offs, scale, E)
    else:
        w13, w2 = layer.mlp._stacked()
        gu = lambda: ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
        act = gu()
        dn = lambda: ops.grouped_gemm_down(act, w2, offs, scale, E)
    for name, fn, flops in (("grouped_gemm_gateup_silu@prefill16k", gu, 2.0 * R * 2 * I * H),
                            ("grouped_gemm_down@prefill16k", dn, 2.0 * R * H * I)):
        us = timed(fn)
        tf = flops / (us * 1e-6) / 1e12
        out[name] = {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(tf / peak, 4), "traffic": None, "kernel_us_mean": round(us, 1),
                     "algorithmic_flops": flops, "flops_model": f"2*R*N*K with R={R} rows (T={T}, k={k}, +shared)"}
    # the grouped expert GEMM as ONE figure: both launches' FLOPs over both launches' time
    a, b = out["grouped_gemm_gateup_silu@prefill16k"], out["grouped_gemm_down@prefill16k"]
    fl, us = a["algorithmic_flops"] + b["algorithmic_flops"], a["kernel_us_mean"] + b["kernel_us_mean"]
    tf = fl / (us * 1e-6) / 1e12
    out["grouped_gemm@prefill16k"] = {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
                                      "frac": round(tf / peak, 4), "traffic": None,
                                      "kernel_us_mean": round(us, 1), "algorithmic_flops": fl,
                                      "flops_model": "gate|up + down launches together"}
    if quant_us is not None:  # the one extra launch of the W8A8 pair: x -> e4m3 rows + per-token scales (16384 x 4096)
        tfq = fl / ((us + quant_us) * 1e-6) / 1e12
        out["grouped_gemm@prefill16k"].update(quantize_x_us=round(quant_us, 1), achieved_incl_quantize_x=round(tfq, 1),
                                              frac_incl_quantize_x=round(tfq / peak, 4))
    return out


def _events_us(fn, iters, warm):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def measure_extras(model, cfg, ops, dev, L, T=16384, S=4096):
    """Prefill side of the same config (T = 16384 = max_num_batched_tokens, 4 prompts x 4096), each entry with its
    steps, time and FLOP model; random data.  Runs after every decode measurement (the prefill step overwrites KV rows)."""
    import torch
    from nanovllm.utils.context import set_context
    out = {}
    nh, nkv, hd, H = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
    Bp = T // S
    mfma = lambda fl, us: {"ms": round(us / 1e3, 4), "tflops": round(fl / (us * 1e-6) / 1e12, 1),
                           "mfma_frac": round(fl / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS, 4)}
    try:
        # (1) varlen causal attention, 4 x 4096 tokens, 96 / 8 heads x 128
        q = torch.randn(T, nh, hd, device=dev).to(torch.bfloat16)
        k = torch.randn(T, nkv, hd, device=dev).to(torch.bfloat16)
        v = torch.randn(T, nkv, hd, device=dev).to(torch.bfloat16)
        cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
        us = _events_us(lambda: ops.varlen_attn_prefill(q, k, v, cu, cu, S, hd ** -0.5), 8, 3)
        fl = 4.0 * Bp * nh * S * S * hd / 2
        out["prefill_attn_4x4096"] = dict(mfma(fl, us), steps=8, flops=fl,
                                          flops_model="4*B*nh*S^2*hd/2: QK^T and PV over the causal half, B=4 S=4096 nh=96 hd=128")
        del q, k, v
        # (2) dense prefill projections on the large-M loop
        a0 = model.model.layers[1].self_attn
        x = torch.randn(T, H, device=dev).to(torch.bfloat16)
        us = _events_us(lambda: ops.gemm_bf16(x, a0.qkv_proj.weight.data, a0.qkv_proj.bias.data), 8, 3)
        fl = 2.0 * T * (nh + 2 * nkv) * hd * H
        out["dense_prefill_qkv"] = dict(mfma(fl, us), steps=8, flops=fl, flops_model="2*T*N*K, T=16384 N=14336 K=4096 (+bias)")
        xo = torch.randn(T, nh * hd, device=dev).to(torch.bfloat16)
        us = _events_us(lambda: ops.gemm_bf16(xo, a0.o_proj.weight.data, None), 8, 3)
        fl = 2.0 * T * H * nh * hd
        out["dense_prefill_o"] = dict(mfma(fl, us), steps=8, flops=fl, flops_model="2*T*N*K, T=16384 N=4096 K=12288")
        del x, xo
        # (3) a whole prefill step of 16384 tokens through 3 MoE decoder layers (eager; K/V written to the paged cache)
        layers = [l for l in model.model.layers if hasattr(l.mlp, "gate")][:3]
        h0 = torch.randn(T, H, device=dev).to(torch.bfloat16)
        pos = torch.arange(S, device=dev).repeat(Bp)
        slots = torch.arange(T, dtype=torch.int32, device=dev)
        set_context(True, cu, cu, S, S, slots, None, None)

        @torch.inference_mode()
        def pstep():
            hs, res = h0, None
            for l in layers:
                hs, res = l(pos, hs, res, defer_combine=True)
            return hs

        us = _events_us(pstep, 3, 1)
        E, kk, I = cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
        fl_layer = (2.0 * T * (nh + 2 * nkv) * hd * H + 4.0 * Bp * nh * S * S * hd / 2 + 2.0 * T * H * nh * hd
                    + 2.0 * T * E * H + 2.0 * T * (kk + 1) * 3 * I * H)
        ms_layer = us / 1e3 / len(layers)
        out["prefill_step_16k"] = {"ms_per_layer": round(ms_layer, 4), "layers": len(layers), "steps": 3, "tokens": T,
                                   "tokens_per_s_at_L": round(T / (ms_layer * 1e-3 * L), 1), "L": L,
                                   "tflops": round(fl_layer / (ms_layer * 1e-3) / 1e12, 1),
                                   "mfma_frac": round(fl_layer / (ms_layer * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                                   "flops_per_layer": fl_layer,
                                   "flops_model": "QKV + causal attention + O + router + (k+1) experts x 3*I*H per token, per MoE layer"}
    except torch.OutOfMemoryError as e:
        out["extras_error"] = repr(e)[:160]
    torch.cuda.empty_cache()
    return out


def measure_fp8(model, cfg, ops, step, B, ctx, L, args, dev):
    """BASELINE config 5 flavour on the headline model: experts re-quantised to OCP fp8 e4m3 (weight-only, per-row
    scale; the bf16 stacks are released), the same decode step captured again, plus the two expert launches alone."""
    import torch
    for l in model.model.layers:
        if hasattr(l.mlp, "gate"):
            l.mlp.expert_weight_dtype = "fp8_e4m3"
            l.mlp.quantize_experts_fp8()
    cfg.expert_weight_dtype = "fp8_e4m3"
    torch.cuda.empty_cache()
    step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    steps = max(5, args.steps // 2)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # twice, the second reading reported: right after the CPU-baseline leg the first isolated measurement of this leg read
    # 117 / 144 / 158 us for the same gate|up launch on three boxes while the captured step around it was stable (r03)
    measure_kernels(model, cfg, ops, B, ctx, 1, max(4, args.kernel_iters // 4), dev, only_experts=True)
    ks = measure_kernels(model, cfg, ops, B, ctx, 1, args.kernel_iters, dev, only_experts=True)
    moe_layer = next(l for l in model.model.layers if hasattr(l.mlp, "gate"))
    pf = measure_prefill_gemm(moe_layer, cfg, ops, 1, dev, iters=6, warm=3, fp8=True)
    try:  # r05: the same pair as W8A8 on the block-scaled fp8 MFMA (opt-in path; parity unpinned)
        pa = measure_prefill_gemm(moe_layer, cfg, ops, 1, dev, iters=6, warm=3, a8=True)
        a8 = {"prefill16k_pair_tflops": pa["grouped_gemm@prefill16k"]["achieved"],
              "prefill16k_pair_frac_of_5pf_fp8_peak": pa["grouped_gemm@prefill16k"]["frac"],
              "prefill16k_pair_us": pa["grouped_gemm@prefill16k"]["kernel_us_mean"],
              "prefill16k_gateup_tflops": pa["grouped_gemm_gateup_silu@prefill16k"]["achieved"],
              "prefill16k_down_tflops": pa["grouped_gemm_down@prefill16k"]["achieved"],
              "quantize_x_us": pa["grouped_gemm@prefill16k"].get("quantize_x_us"),
              "prefill16k_pair_tflops_incl_quantize_x": pa["grouped_gemm@prefill16k"].get("achieved_incl_quantize_x"),
              "prefill16k_model": "W8A8, e4m3 x e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4: x quantised per token (one extra launch, "
                                  "timed), act quantised per (row, 32 columns) with E8M0 scales by the gate|up epilogue, the down "
                                  "projection takes those block scales inside the MFMA; peak = 5 PFLOP/s dense fp8"}
    except Exception as e:  # noqa: BLE001
        a8 = {"error": repr(e)[:200]}
    # ... and the whole 16384-token prefill step through 3 MoE layers (as extra.prefill_step_16k) with fp8 experts: weight-only
    # against W8A8 (LAST: it overwrites K/V rows)
    try:
        from nanovllm.utils.context import set_context
        T, S = 16384, 4096
        layers = [l for l in model.model.layers if hasattr(l.mlp, "gate")][:3]
        h0 = torch.randn(T, cfg.hidden_size, device=dev).to(torch.bfloat16)
        pos = torch.arange(S, device=dev).repeat(T // S)
        cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
        set_context(True, cu, cu, S, S, torch.arange(T, dtype=torch.int32, device=dev), None, None)

        @torch.inference_mode()
        def pstep():
            hs, res = h0, None
            for l in layers:
                hs, res = l(pos, hs, res, defer_combine=True)
            return hs

        ms = {}
        for tag, act_dtype in (("weight_only", None), ("w8a8", "fp8_e4m3")):
            for l in layers:
                l.mlp.expert_activation_dtype = act_dtype
            ms[tag] = round(_events_us(pstep, 3, 1) / 1e3 / len(layers), 4)
        for l in layers:
            l.mlp.expert_activation_dtype = None
        a8["prefill_step_16k_ms_per_layer"] = ms["w8a8"]
        a8["prefill_step_16k_ms_per_layer_weight_only"] = ms["weight_only"]
        del h0
    except Exception as e:  # noqa: BLE001
        a8["prefill_step_error"] = repr(e)[:200]
    return {"workload": f"same model and decode step (bs={B} ctx={ctx} L={L}), experts as fp8 e4m3 weight-only",
            # the pair on the fp8 MFMA (W8A8, opt-in) is THE prefill figure of this entry since r05; the weight-only pair (bf16
            # MFMA, the parity path) keeps its own keys
            "prefill16k_pair_tflops": a8.get("prefill16k_pair_tflops", pf["grouped_gemm@prefill16k"]["achieved"]),
            "prefill16k_pair_frac_of_5pf_fp8_peak": a8.get("prefill16k_pair_frac_of_5pf_fp8_peak"),
            "prefill16k_pair_us": a8.get("prefill16k_pair_us", pf["grouped_gemm@prefill16k"]["kernel_us_mean"]),
            "w8a8": a8,
            "weight_only": {"prefill16k_gateup_tflops": pf["grouped_gemm_gateup_silu@prefill16k"]["achieved"],
                            "prefill16k_down_tflops": pf["grouped_gemm_down@prefill16k"]["achieved"],
                            "prefill16k_pair_tflops": pf["grouped_gemm@prefill16k"]["achieved"],
                            "prefill16k_pair_us": pf["grouped_gemm@prefill16k"]["kernel_us_mean"]},
            "prefill16k_gateup_tflops": pf["grouped_gemm_gateup_silu@prefill16k"]["achieved"],
            "prefill16k_down_tflops": pf["grouped_gemm_down@prefill16k"]["achieved"],
            "prefill16k_model": "weight_only: grouped gate|up + down at T=16384 with the fp8 bytes as the tiled kernel's weight "
                                "operand (bf16 MFMA, fp32 accumulate, per-row scale in the epilogue); w8a8: see there",
            "tokens_per_s": round(B / dt, 2), "ms_per_step": round(dt * 1e3, 4), "ms_per_layer": round(dt * 1e3 / L, 4),
            "steps": steps, "hip_graph": True,
            "gateup_hbm_frac": ks["grouped_gemm_gateup_silu"]["frac"], "gateup_us": ks["grouped_gemm_gateup_silu"]["kernel_us_mean"],
            "down_hbm_frac": ks["grouped_gemm_down"]["frac"], "down_us": ks["grouped_gemm_down"]["kernel_us_mean"],
            "bytes_model": "touched groups x 3*I*H x 1 B of fp8 weights per launch pair"}


def cpu_baseline(model, cfg, kv, block_tables, context_lens, positions, slot_mapping, B, L):
    """The reference's eager algorithm (CPU oracle) for ONE MoE decoder layer at this workload's
    shapes, timed on the host cores; tokens/s is extrapolated to L layers."""
    import torch
    from oracle import glm4_moe_oracle as O
    # BASELINE.md §4 asks for os.cpu_count() threads.  A 1-GPU box owns a 16-core SHARE of a host that reports 256:
    # 256 threads oversubscribe that share and run the same GEMMs 100-300x slower (r01 probe, tools/cpu_probe.py), so
    # the thread count is capped at the share; both numbers are reported.
    host_cpus = os.cpu_count() or 1
    ncores = min(16, host_cpus)
    torch.set_num_threads(ncores)
    layer = next(l for l in model.model.layers if hasattr(l.mlp, "gate"))
    li = list(model.model.layers).index(layer)
    cpu = lambda t: t.detach().to("cpu")
    from oracle import from_product
    lw = from_product.layer_weights_from_module(layer)  # the bytes the GPU ran on (fp8 experts: the quantised bytes + scales)
    kc, vc = cpu(kv[0, li]), cpu(kv[1, li])
    c = {k: getattr(cfg, k) for k in ("num_attention_heads", "num_key_value_heads", "head_dim", "rms_norm_eps",
                                      "num_experts_per_tok", "routed_scaling_factor", "partial_rotary_factor")}
    cs = O.build_cos_sin_cache(int(cfg.head_dim * cfg.partial_rotary_factor), cfg.max_position_embeddings, cfg.rope_theta)
    meta = O.AttnMeta(False, slot_mapping=cpu(slot_mapping), context_lens=cpu(context_lens), block_tables=cpu(block_tables))
    h = torch.randn(B, cfg.hidden_size).to(torch.bfloat16)
    r = torch.randn(B, cfg.hidden_size).to(torch.bfloat16)
    pos = cpu(positions)
    times = []
    with torch.inference_mode():
        for it in range(4):  # 1 warm-up + 3 timed, ~3 s each
            t0 = time.perf_counter()
            O.decoder_layer(pos, h, r, lw, c, cs, kc, vc, meta)
            times.append(time.perf_counter() - t0)
            if sum(times) > 60:
                break
    timed = times[1:] if len(times) > 1 else times
    t_layer = sorted(timed)[len(timed) // 2]
    return {"value": round(B / (L * t_layer), 3), "unit": "tokens/s", "cores": ncores, "host_cpu_count": host_cpus,
            "kind": "port",
            "sample": f"one MoE decoder layer, decode step B={B} ctx={int(context_lens[0])}, torch CPU bf16, "
                      f"{len(timed)} timed run(s) after 1 warm-up, median {t_layer * 1e3:.0f} ms/layer-step; "
                      f"tokens/s = B / (L={L} x t_layer)",
            "ms_per_layer_step": round(t_layer * 1e3, 1)}


def cpu_baseline_c1(ops, dev, ncores):
    """BASELINE configs[0] — ONE GLM-4-MoE decoder layer, 2 experts top-1, bs=1 seq=128, CPU eager (the plumbing config;
    reference path models/glm4_moe/decode_layer.py:79-100, its smoke examples/decode_layer.py:7-55) — timed on the host
    cores with the CPU oracle on the committed fixture minted from the reference (tests/golden/layer_c1.safetensors:
    H=128, 4/2 heads x 128, I=64), bf16 regime and fp32 modules, next to the HIP path on the same inputs (checked
    against the fixture before it is timed)."""
    import torch
    from oracle import glm4_moe_oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import build_product_layer, layer_weights_from_golden, load_golden
    from nanovllm.utils.context import reset_context, set_context
    t, cfg = load_golden("layer_c1")
    lw = layer_weights_from_golden(t, cfg)
    N = t["h"].shape[0]
    cu = t["cu_seqlens"]
    meta = O.AttnMeta(True, cu_seqlens_q=cu, cu_seqlens_k=cu)

    def med(fn, n=7):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[n // 2]

    with torch.inference_mode():
        out = O.decoder_layer(t["positions"], t["h"], None, lw, cfg, t["cos_sin"], None, None, meta)[0]
        # (bit equality with the reference's vector is asserted in the build container, tests/test_oracle_vs_golden.py; the
        # box's thread count changes the CPU GEMM's summation order by single bf16 ulps)
        o, g = out.float(), t["out_first"].float()
        assert bool(((o - g).abs() <= 2 ** -6 * g.abs() + 2 ** -6 * float(g.abs().max())).all()), "oracle drifted from the C1 golden vector"
        t_bf16 = med(lambda: O.decoder_layer(t["positions"], t["h"], None, lw, cfg, t["cos_sin"], None, None, meta))
        lw32, h32 = lw.to(torch.float32), t["h"].float()
        t_f32 = med(lambda: O.decoder_layer(t["positions"], h32, None, lw32, cfg, t["cos_sin"], None, None, meta))
    # the HIP path on the same inputs (cache-less prompt, first-layer form), checked, then timed with events
    layer = build_product_layer(t, cfg, "model.layers.1", device=dev)
    cud, posd, hd_ = cu.to(dev), t["positions"].to(dev), t["h"].to(dev)
    set_context(True, cud, cud, N, N, None, None, None)

    @torch.inference_mode()
    def gpu_step():
        return layer(posd, hd_, None)[0]

    got = gpu_step().cpu().float()
    ref = t["out_first"].float()
    frac_ok = float(((got - ref).abs() <= 2 ** -6 * ref.abs() + 2 ** -6 * float(ref.abs().max())).float().mean())
    us = _events_us(gpu_step, 20, 3)
    reset_context()
    return {"workload": "BASELINE configs[0]: one MoE decoder layer, E=2 top-1, bs=1 seq=128 (fixture dims H=128, 4/2 heads x 128, I=64)",
            "cpu_ms_bf16": round(t_bf16 * 1e3, 3), "cpu_ms_fp32": round(t_f32 * 1e3, 3), "cores": ncores, "kind": "port",
            "cpu_tokens_per_s_bf16": round(N / t_bf16, 1), "sample": "7 timed runs after 1 warm-up, median",
            "hip_us": round(us, 1), "hip_tokens_per_s": round(N / (us * 1e-6), 1),
            "hip_rows_within_golden_tolerance": round(frac_ok, 4)}


if __name__ == "__main__":
    main()
