"""Per-GPU model runner — reference: engine/model_runner.py:16-258.

One process per GPU.  Rank 0 lives in the user's process and drives ranks 1..tp-1 (spawned by
LLMEngine) through `torch.distributed` broadcasts on a gloo side group (the reference uses a POSIX
shm segment + mp.Event + pickle, model_runner.py:51-91); the data plane is RCCL ("nccl").
The decode step is captured in HIP graphs per batch-size bucket — possible for GLM-4-MoE here
because the MoE block has no host sync (the reference has to run it eagerly)."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from nanovllm.config import Config
from nanovllm.engine.sequence import Sequence
from nanovllm.layers.sampler import Sampler
from nanovllm.models.glm4_moe.model import Glm4MoeForCausalLM
from nanovllm.utils.context import get_context, reset_context, set_context


class ModelRunner:

    def __init__(self, config: Config, rank: int, master_port: int = 2333):
        self.config = config
        hf = config.hf_config
        self.block_size = config.kvcache_block_size
        self.enforce_eager = config.enforce_eager
        self.world_size = config.tensor_parallel_size
        self.rank = rank
        self.ctrl_group = None
        self.graph_note = None
        if self.world_size > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(master_port))
            capture_ok = True
            if not self.enforce_eager:
                # may the decode step be captured WITH its RCCL all-reduces?  Asked in a throwaway child per rank before
                # this rank's communicator exists (a failed capture leaves a process unusable and its peers blocked)
                from nanovllm.distributed import capture_probe
                env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(self.world_size))
                if int(getattr(hf, "expert_parallel_size", 1) or 1) > 1:
                    # the captured step then holds an all-to-all and an all-gather as well: probe those too
                    env["GLM_PROBE_EP"] = "1"
                capture_ok, self.graph_note = capture_probe.probe(env)
            torch.cuda.set_device(rank)
            dist.init_process_group("nccl", world_size=self.world_size, rank=rank,
                                    device_id=torch.device("cuda", rank))
            self.ctrl_group = dist.new_group(backend="gloo")
            if not self.enforce_eager:
                from nanovllm.distributed import capture_probe
                if not capture_probe.agree(capture_ok, torch.device("cuda", rank)):
                    self.enforce_eager = True  # on EVERY rank
                    self.graph_note = self.graph_note or "capture probe failed on another rank"
        torch.cuda.set_device(rank)
        self.device = torch.device("cuda", rank)
        self.ep = int(getattr(hf, "expert_parallel_size", 1) or 1) > 1
        if self.ep:
            # tight all-to-all slots (2x the mean rows per slot unless the config says otherwise; 0 = worst case): this
            # runner is the caller that checks the plan's overflow word after every step and redoes an overflowed one
            from nanovllm.distributed.expert_parallel import EPExperts
            EPExperts.set_capacity_factor(getattr(hf, "expert_parallel_capacity_factor", 2.0))
        old_dtype = torch.get_default_dtype()
        torch.set_default_dtype(hf.torch_dtype)
        try:
            with torch.device(self.device):
                model_type = getattr(hf, "model_type", "glm4_moe")
                if model_type != "glm4_moe":
                    raise NotImplementedError(f"model_type {model_type!r}: only glm4_moe is built here")
                self.model = Glm4MoeForCausalLM(hf)
                self.model.load_weights(config.model)
                self.sampler = Sampler()
                self.warmup_model()
                self.reserve_workspaces()
                self.allocate_kv_cache()
                if not self.enforce_eager:
                    self.capture_graphs()
        finally:
            torch.set_default_dtype(old_dtype)

    # ------------------------------------------------------------------ control plane
    def exit(self):
        if not self.enforce_eager:
            self.graphs = {}
        torch.cuda.synchronize()
        if self.world_size > 1:
            dist.barrier()
            dist.destroy_process_group()

    def loop(self):
        """Worker ranks: execute what rank 0 broadcasts until "exit"."""
        while True:
            msg = [None]
            dist.broadcast_object_list(msg, src=0, group=self.ctrl_group)
            name, args = msg[0]
            self.call(name, *args)
            if name == "exit":
                break

    def call(self, name: str, *args):
        if self.world_size > 1 and self.rank == 0:
            dist.broadcast_object_list([(name, args)], src=0, group=self.ctrl_group)
        return getattr(self, name)(*args)

    # ------------------------------------------------------------------ memory
    def warmup_model(self):
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        cfg = self.config
        n = max(1, min(cfg.max_num_batched_tokens // cfg.max_model_len, cfg.max_num_seqs))
        self.run([Sequence([0] * cfg.max_model_len) for _ in range(n)], True)
        torch.cuda.empty_cache()

    def reserve_workspaces(self):
        """Persistent scratch that must exist before the KV cache takes the rest of the memory: the split-KV partials
        of the decode attention, sized for every batch size up to max_num_seqs (a few MB: no splits at all once
        batch x kv-heads covers the chip), and the router launch's workspace.  Both are keyed by device, so the graphs
        captured later (on torch's capture stream) use exactly these buffers; growing one during a capture raises."""
        from nanovllm import hip_ops
        cfg, hf = self.config, self.config.hf_config
        nh = hf.num_attention_heads // self.world_size
        nkv = max(1, hf.num_key_value_heads // self.world_size)
        max_blocks = (cfg.max_model_len + self.block_size - 1) // self.block_size
        if hf.head_dim == 128:
            hip_ops.reserve_attn_workspace(self.device, min(cfg.max_num_seqs, 512), nh, nkv, hf.head_dim,
                                           self.block_size, max_blocks)
        if getattr(hf, "n_routed_experts", None):  # arrival counters + logits scratch of the fused norm + router launch
            hip_ops.reserve_router_workspace(self.device, min(cfg.max_num_seqs, 512), hf.n_routed_experts)

    def allocate_kv_cache(self):
        cfg, hf = self.config, self.config.hf_config
        n_layers = sum(1 for m in self.model.modules() if hasattr(m, "k_cache") and hasattr(m, "v_cache"))
        nkv = max(1, hf.num_key_value_heads // self.world_size)
        block_bytes = 2 * n_layers * self.block_size * nkv * hf.head_dim * hf.torch_dtype.itemsize
        if cfg.num_kvcache_blocks <= 0:
            free, total = torch.cuda.mem_get_info()
            stats = torch.cuda.memory_stats()
            peak, cur = stats["allocated_bytes.all.peak"], stats["allocated_bytes.all.current"]
            budget = int(total * cfg.gpu_memory_utilization) - (total - free) - (peak - cur)
            cfg.num_kvcache_blocks = budget // block_bytes
        assert cfg.num_kvcache_blocks > 0, "no memory left for the KV cache"
        self.kv_cache = torch.empty(2, n_layers, cfg.num_kvcache_blocks, self.block_size, nkv, hf.head_dim,
                                    dtype=hf.torch_dtype, device=self.device)
        lid = 0
        for m in self.model.modules():
            if hasattr(m, "k_cache") and hasattr(m, "v_cache"):
                m.k_cache, m.v_cache = self.kv_cache[0, lid], self.kv_cache[1, lid]
                lid += 1

    # ------------------------------------------------------------------ step inputs
    def _dev(self, data, dtype):
        return torch.tensor(data, dtype=dtype, pin_memory=True).to(self.device, non_blocking=True)

    def prepare_block_tables(self, seqs):
        width = max(len(s.block_table) for s in seqs)
        return self._dev([s.block_table + [-1] * (width - len(s.block_table)) for s in seqs], torch.int32)

    def prepare_prefill(self, seqs):
        ids, pos, slots, cu_q, cu_k = [], [], [], [0], [0]
        max_q = max_k = 0
        for s in seqs:
            n, cached = len(s), s.num_cached_tokens
            ids.extend(s[cached:])
            pos.extend(range(cached, n))
            cu_q.append(cu_q[-1] + n - cached)
            cu_k.append(cu_k[-1] + n)
            max_q, max_k = max(max_q, n - cached), max(max_k, n)
            if not s.block_table:  # warm-up: nothing to store
                continue
            for i in range(s.num_cached_blocks, s.num_blocks):
                start = s.block_table[i] * self.block_size
                count = self.block_size if i != s.num_blocks - 1 else s.last_block_num_tokens
                slots.extend(range(start, start + count))
        bt = self.prepare_block_tables(seqs) if cu_k[-1] > cu_q[-1] else None  # prefix-cache hit
        set_context(True, self._dev(cu_q, torch.int32), self._dev(cu_k, torch.int32), max_q, max_k,
                    self._dev(slots, torch.int32), None, bt)
        return self._dev(ids, torch.int64), self._dev(pos, torch.int64)

    def prepare_decode(self, seqs):
        ids = [s.last_token for s in seqs]
        pos = [len(s) - 1 for s in seqs]
        lens = [len(s) for s in seqs]
        slots = [s.block_table[-1] * self.block_size + s.last_block_num_tokens - 1 for s in seqs]
        set_context(False, slot_mapping=self._dev(slots, torch.int32), context_lens=self._dev(lens, torch.int32),
                    block_tables=self.prepare_block_tables(seqs))
        return self._dev(ids, torch.int64), self._dev(pos, torch.int64)

    # ------------------------------------------------------------------ execution
    @torch.inference_mode()
    def run_model(self, input_ids, positions, is_prefill, eager: bool = False):
        bs = input_ids.size(0)
        if is_prefill or self.enforce_eager or eager or bs > self.graph_bs[-1]:
            return self.model.compute_logits(self.model(input_ids, positions))
        ctx = get_context()
        bucket = next(b for b in self.graph_bs if b >= bs)
        gv = self.graph_vars
        gv["input_ids"][:bs] = input_ids
        gv["positions"][:bs] = positions
        gv["slot_mapping"].fill_(-1)
        gv["slot_mapping"][:bs] = ctx.slot_mapping
        gv["context_lens"].zero_()
        gv["context_lens"][:bs] = ctx.context_lens
        gv["block_tables"][:bs, : ctx.block_tables.size(1)] = ctx.block_tables
        self.graphs[bucket].replay()
        return self.model.compute_logits(gv["outputs"][:bs])

    def run(self, seqs: list[Sequence], is_prefill: bool):
        input_ids, positions = self.prepare_prefill(seqs) if is_prefill else self.prepare_decode(seqs)
        temps = self._dev([s.temperature for s in seqs], torch.float32) if self.rank == 0 else None
        logits = self.run_model(input_ids, positions, is_prefill)
        tokens = self.sampler(logits, temps).tolist() if self.rank == 0 else None
        if self.ep:
            # expert parallelism with tight slots: did a (source, destination) slot overflow in this step?  The word is
            # computed from the replicated routing, so every rank reads the same value and takes the same branch.  An
            # overflowed step dropped rows: redo it eagerly with exact (variable) splits — K/V rows are rewritten with
            # the same values, nothing else was kept from the first attempt.
            from nanovllm.distributed.expert_parallel import EPExperts
            if EPExperts.overflowed():
                EPExperts.force_varlen(True)
                try:
                    logits = self.run_model(input_ids, positions, is_prefill, eager=True)
                    tokens = self.sampler(logits, temps).tolist() if self.rank == 0 else None
                finally:
                    EPExperts.force_varlen(False)
                self.ep_redone_steps = getattr(self, "ep_redone_steps", 0) + 1
        from nanovllm import hip_ops
        hip_ops.check_router_tail(self.device)  # a bounded in-launch wait that expired (cannot while the GPU makes progress)
        hip_ops.check_decode_engine(self.device)
        if self.world_size > 1:
            # a one-shot all-reduce that gave up waiting poisoned its output and set a status word: surface it here, where
            # rank 0 has just synchronised on the tokens anyway (no-op on RCCL)
            from nanovllm.distributed.parallel_state import get_tp_group
            get_tp_group().check()
        reset_context()
        return tokens

    @torch.inference_mode()
    def capture_graphs(self):
        cfg, hf = self.config, self.config.hf_config
        max_bs = min(cfg.max_num_seqs, 512)
        max_blocks = (cfg.max_model_len + self.block_size - 1) // self.block_size
        dev = self.device
        gv = dict(input_ids=torch.zeros(max_bs, dtype=torch.int64, device=dev),
                  positions=torch.zeros(max_bs, dtype=torch.int64, device=dev),
                  slot_mapping=torch.full((max_bs,), -1, dtype=torch.int32, device=dev),
                  context_lens=torch.zeros(max_bs, dtype=torch.int32, device=dev),
                  block_tables=torch.zeros(max_bs, max_blocks, dtype=torch.int32, device=dev),
                  outputs=torch.zeros(max_bs, hf.hidden_size, dtype=hf.torch_dtype, device=dev))
        self.graph_bs = sorted({b for b in (1, 2, 4, 8) if b <= max_bs} | set(range(16, max_bs + 1, 16)) | {max_bs})
        self.graphs, pool = {}, None
        for bs in reversed(self.graph_bs):
            set_context(False, slot_mapping=gv["slot_mapping"][:bs], context_lens=gv["context_lens"][:bs],
                        block_tables=gv["block_tables"][:bs])
            gv["outputs"][:bs] = self.model(gv["input_ids"][:bs], gv["positions"][:bs])  # warm-up
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool):
                gv["outputs"][:bs] = self.model(gv["input_ids"][:bs], gv["positions"][:bs])
            pool = pool or g.pool()
            self.graphs[bs] = g
            torch.cuda.synchronize()
            reset_context()
        if self.ep:
            # the warm-ups and captures above ran on all-zero token ids: every row routes alike, which may have set the
            # sticky overflow word — the first real step must not be redone for it
            from nanovllm.distributed.expert_parallel import EPExperts
            EPExperts.clear_overflow()
        self.graph_vars = gv
