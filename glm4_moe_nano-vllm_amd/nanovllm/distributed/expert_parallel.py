"""Expert parallelism for the GLM-4-MoE block (BASELINE config 4: TP=8 over xGMI with expert-parallel all-to-all).

No reference counterpart runs: the reference has no all-to-all call site, only dead `dispatch` / `combine` stubs
(distributed/parallel_state_ref.py:792-810).  Parity is therefore defined as "EP = k output == TP = 1 output"
(SURVEY.md §8e): the routing, every expert row and the combine order (ascending expert id, one bf16 rounding per add,
moe.py:133-171) are the ones of the single-GPU path; only WHERE a row is computed changes.

Layout, ep ranks (= the TP group):
  * rank r owns the routed experts [r*E/ep, (r+1)*E/ep) with their FULL intermediate size (not TP-sharded) and a replica
    of the shared expert;
  * the hidden states are replicated after the attention's all-reduce; rank r takes the token slice
    [r*ceil(T/ep), ...), and for its tokens
      1. buckets its (token, expert) pairs by expert (glm_moe_align — expert order is also destination-rank order) and
         packs the rows (glm_gather_rows);
      2. all-gathers the per-expert pair counts (E int32 per rank), all-to-all-v's the rows to the experts' owners;
      3. the owner runs its experts over the rows it received, grouped by expert through a gather index
         (glm_grouped_gemm_gateup_silu / glm_grouped_gemm_down, unscaled), un-permutes them (glm_scatter_rows) and
         all-to-all-v's them back;
      4. the source applies the routing weights and combines in ascending expert id (glm_moe_combine_scaled) and adds
         its shared-expert rows; the slices are all-gathered so that every rank has [T, H] again for the next
         (tensor-parallel) attention.
  xGMI is point-to-point: at B=256, k=8 a rank sends 256 pairs x 8 KiB = 2 MiB spread over 7 peers — one hop, all links.

The forward is written as a GENERATOR that yields at every collective.  `run` drives it with torch.distributed (RCCL,
or gloo in the CPU tests); `run_emulated` drives the generators of several logical ranks in lockstep inside one process
(the one-GPU rehearsal and the "EP=2 == TP=1" test).

Two dispatch forms, same arithmetic, same results:
  * FIXED CAPACITY (r03; decode steps of up to FIXED_MAX_TOKENS tokens and every captured step; graph-capturable): the
    routing is replicated, so every rank can compute the whole exchange plan on the device (glm_ep_plan: send slots,
    receive-side expert offsets and gather list) — no count all-gather, no host read.  The all-to-all moves equal,
    capacity-sized slots ([ep][cap] rows); the grouped GEMMs run over the capacity-sized receive buffer with
    expert_offsets saying which rows are real.  Three collectives per MoE block (all-to-all, all-to-all, all-gather), all
    with static shapes.  cap = the worst case (rows per token slice x min(k, experts per rank), overflow impossible) or,
    when the caller opted in with `EPExperts.set_capacity_factor(f)`, f x the mean with the plan's `overflow` word
    sending an overflowing step to the variable-split form (see the class).
  * VARIABLE SPLITS (`forward_varlen`: prefill-sized steps, the redo of an overflowed step, option ep_fixed = 0): all-to-all-v
    with exact row counts; the split sizes are host integers, so this form reads the count matrix once per MoE block and
    runs eagerly.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def token_slice(T: int, ep: int, rank: int) -> tuple[int, int]:
    per = (T + ep - 1) // ep
    lo = min(T, rank * per)
    return lo, min(T, lo + per)


class EPExperts:
    """One rank's side of the expert-parallel MoE block.  `ops` is nanovllm.hip_ops (the CPU tests inject a stand-in
    with the reference arithmetic on CPU tensors); `w13`, `w2` are this rank's [E/ep, 2I, H] / [E/ep, H, I] stacks, `shared` a callable
    for the (replicated) shared expert."""

    def __init__(self, ops, ep_size: int, rank: int, n_experts: int, top_k: int, w13, w2, shared, fp8=None):
        """fp8 (r04): (w13_q, w13_scale, w2_q, w2_scale) of this rank's experts — fp8 e4m3 weight-only stacks with per-row
        fp32 scales (hip_ops.quantize_fp8_rowwise); the grouped GEMMs then stream those (w13 / w2 may be None)."""
        assert n_experts % ep_size == 0, "experts must divide over the expert-parallel ranks"
        self.ops, self.ep, self.rank, self.E, self.k = ops, ep_size, rank, n_experts, top_k
        self.El = n_experts // ep_size
        self.w13, self.w2, self.shared, self.fp8 = w13, w2, shared, fp8

    def _experts(self, rows, offs, gidx, n_rows: int, expected: int | None):
        """This rank's experts over `rows` (grouped by `offs`, gathered through `gidx`): gate|up + SiLU*mul, then the
        UNSCALED down projection (the routing weight is applied at the token's home rank).  expected: the fixed-capacity
        form's row-count hint (None = exact rows)."""
        ops = self.ops
        if self.fp8 is not None:
            # the fp8 launchers walk the groups through `offs` like the bf16 ones: rows past offs[El] are not touched
            w13_q, w13_s, w2_q, w2_s = self.fp8
            act = ops.grouped_gemm_gateup_silu_fp8w(rows, w13_q, w13_s, offs, gidx, n_rows)
            return ops.grouped_gemm_down_fp8w(act, w2_q, w2_s, offs, None, 0)
        if expected is not None:
            act = ops.grouped_gemm_gateup_silu_cap(rows, self.w13, offs, gidx, n_rows, expected)
            return ops.grouped_gemm_down_cap(act, self.w2, offs, expected)
        act = ops.grouped_gemm_gateup_silu(rows, self.w13, offs, gidx, n_rows)
        return ops.grouped_gemm_down(act, self.w2, offs, None, 0)

    FIXED_MAX_SLICE_PAIRS = 16384  # glm_ep_plan keeps a token slice's pairs in LDS
    # The fixed-capacity form exists for DECODE steps (graph buckets): every buffer has ep*cap rows whatever the routing.
    # A prefill-sized step (T = 16384 at ep = 8: five [131072, 4096] tensors of 1 GiB each at the worst-case capacity)
    # takes the variable-split form, which moves exactly the real rows; only a step that is being captured in a HIP graph
    # may not (a host read cannot be captured).
    FIXED_MAX_TOKENS = 512

    # Tight capacity (r04).  `capacity_factor` f: cap = ceil(f x mean rows per (source, destination) slot) instead of the
    # worst case (at B = 256, ep = 8, k = 8: 64 rows per slot instead of 256 => 4 MB per direction per rank instead of
    # 16 MB).  glm_ep_plan reports a slot that does not fit through its `overflow` word (the SAME value on every rank: the
    # plan is computed from the replicated routing for every (source, destination) pair) and drops the same rows on both
    # sides, so nothing is out of bounds — but the step's result is then wrong and has to be redone.  The word is kept in
    # a sticky device flag; the caller that opted in (`ModelRunner.run`, bench.py, the tests) reads it where it
    # synchronises anyway (`overflowed()`), and re-runs the step with `force_varlen(True)`: a decode step is idempotent
    # (its K/V rows are rewritten with the same values).  None (the default) = worst-case capacity, overflow impossible.
    capacity_factor: float | None = None
    _force_varlen = False
    _flags: dict = {}  # device -> sticky overflow word (int32[1]); created by the first eager step on that device

    @classmethod
    def set_capacity_factor(cls, f: float | None) -> None:
        cls.capacity_factor = None if f is None or f <= 0 else float(f)

    @classmethod
    def force_varlen(cls, on: bool) -> None:
        cls._force_varlen = bool(on)

    @classmethod
    def overflowed(cls, reset: bool = True) -> int:
        """Largest (source, destination) row count that exceeded its slot since the last call, 0 if none (one host read
        per flag: call it where the step is synchronised anyway)."""
        worst = 0
        for f in cls._flags.values():
            v = int(f.item())
            if v:
                worst = max(worst, v)
                if reset:
                    f.zero_()
        return worst

    # below this mean load per (source, destination) slot the factor is NOT applied: the count of a slot is a sum of
    # `per` small integers, and at decode sizes it does not concentrate — simulated under uniform routing, f = 2 overflowed in
    # 47 % of the steps at B = 1, 99.6 % at B = 8, 56 % at B = 32 and ~0 % only from B = 128, and the padded rows
    # of a graph bucket all route alike.  There the worst case costs little anyway (B = 32, ep = 8: 32 rows per slot).
    CAPACITY_MIN_MEAN = 16.0

    def capacity(self, T: int) -> int:
        """Rows per (source, destination) slot.  Worst case: every token of a slice sends at most min(k, experts per
        rank) rows to one rank — can never overflow.  With a capacity factor f, and only when the mean slot load is at
        least CAPACITY_MIN_MEAN: max(f x mean, mean + 4 sqrt(mean)) (the second term keeps a small f from cutting into
        the binomial spread), capped by the worst case."""
        per = (T + self.ep - 1) // self.ep
        worst = max(1, per * min(self.k, self.El))
        f = type(self).capacity_factor
        if f is None:
            return worst
        mean = per * self.k / self.ep
        if mean < self.CAPACITY_MIN_MEAN:
            return worst
        want = max(f * mean, mean + 4.0 * mean ** 0.5)
        return max(1, min(worst, int(-(-want // 1))))

    @classmethod
    def clear_overflow(cls) -> None:
        """Forget what the sticky words hold (after warm-ups / graph capture on dummy inputs, whose rows all route alike)."""
        for f in cls._flags.values():
            f.zero_()

    def _opt_ep_fixed(self) -> bool:
        """Option "ep_fixed" of the real ops module (nanovllm.hip_ops, read once from GLM_EP_FIXED when the library is
        loaded); the CPU tests' stand-in ops module carries its own `ep_fixed` attribute."""
        if hasattr(self.ops, "opt"):
            return self.ops.opt("ep_fixed") != 0
        return bool(getattr(self.ops, "ep_fixed", True))

    def _capturing(self, h) -> bool:
        return bool(h.is_cuda and torch.cuda.is_current_stream_capturing())

    def takes_fixed(self, T: int, capturing: bool = False) -> bool:
        """Which dispatch form a step of T tokens takes (pinned by tests/test_ep_gloo_cpu.py)."""
        per = (T + self.ep - 1) // self.ep
        fits = (hasattr(self.ops, "ep_plan") and per * self.k <= self.FIXED_MAX_SLICE_PAIRS and self.ep * self.E <= 2048)
        if capturing:
            if not fits:
                raise RuntimeError(f"expert-parallel step of {T} tokens cannot be captured (fixed-capacity plan limits)")
            return True
        return (fits and self._opt_ep_fixed() and not type(self)._force_varlen
                and T <= self.FIXED_MAX_TOKENS)

    def forward(self, h: torch.Tensor, topk_idx: torch.Tensor, topk_w: torch.Tensor):
        """Generator.  h [T, H] (replicated), routing of ALL T tokens (replicated, identical on every rank).
        Yields (op, payload...) tuples at the collectives, finally returns the full [T, H] output."""
        fixed = self.takes_fixed(h.shape[0], self._capturing(h))
        return (yield from (self.forward_fixed if fixed else self.forward_varlen)(h, topk_idx, topk_w))

    def _note_overflow(self, ovf: torch.Tensor) -> None:
        if type(self).capacity_factor is None:
            return  # worst-case capacity: the word is always 0
        flags = type(self)._flags
        flag = flags.get(ovf.device)
        if flag is None:
            if self._capturing(ovf):
                raise RuntimeError("expert-parallel overflow flag missing while a HIP graph is being captured: run one eager "
                                   "step with the capacity factor set before capturing")
            flag = flags[ovf.device] = torch.zeros(1, dtype=torch.int32, device=ovf.device)
        torch.maximum(flag, ovf.reshape(1).to(torch.int32), out=flag)

    def forward_fixed(self, h: torch.Tensor, topk_idx: torch.Tensor, topk_w: torch.Tensor):
        """Fixed-capacity dispatch: no host synchronisation anywhere (HIP-graph capturable with captured collectives)."""
        ops, ep, E, El, k = self.ops, self.ep, self.E, self.El, self.k
        T, H = h.shape
        lo, hi = token_slice(T, ep, self.rank)
        Tl = hi - lo
        cap = self.capacity(T)
        send_gather, slot_scale, pair_slot, recv_offs, recv_gidx, overflow = ops.ep_plan(
            topk_idx.contiguous(), topk_w.contiguous(), E, ep, self.rank, cap)
        self._note_overflow(overflow)
        # 1. my pairs' rows into their slots (unused slot rows stay uninitialised: nobody reads them)
        send = ops.gather_rows(h, send_gather)                                  # [ep * cap, H]
        recv = yield ("all_to_all_fixed", send, cap)                            # slot s = the rows rank s sent me
        # 2. my experts over the received rows, in expert order through the gather list; rows beyond recv_offs[El] are
        #    not touched.  The kernel configuration is chosen for the expected row count (T k / ep), not for the capacity.
        expected = max(1, (T * k + ep - 1) // ep)
        y = self._experts(recv, recv_offs, recv_gidx, ep * cap, expected)       # unscaled: the weight is applied at home
        back = ops.scatter_rows(y, recv_gidx, ep * cap, no_init=True)           # expert order -> arrival slots
        ret = yield ("all_to_all_fixed", back, cap)                             # slot d = my rows back from rank d
        # 3. routing weights, ascending-expert combine, shared expert; then everyone gets every slice
        if Tl > 0:
            out_l = ops.moe_combine_scaled(ret, pair_slot, slot_scale, self.shared(h[lo:hi]), Tl, k)
        else:
            out_l = h.new_zeros((0, H))
        sizes = [token_slice(T, ep, r)[1] - token_slice(T, ep, r)[0] for r in range(ep)]
        out = yield ("all_gather_rows", out_l, sizes)
        return out

    def forward_varlen(self, h: torch.Tensor, topk_idx: torch.Tensor, topk_w: torch.Tensor):
        """Variable-split dispatch (exact row counts; one host read of the count matrix per MoE block, eager only)."""
        ops, ep, E, El, k = self.ops, self.ep, self.E, self.El, self.k
        T, H = h.shape
        lo, hi = token_slice(T, ep, self.rank)
        Tl = hi - lo
        hl, idx, w = h[lo:hi], topk_idx[lo:hi].contiguous(), topk_w[lo:hi].contiguous()
        # 1. bucket + pack this rank's pairs (rows sorted by expert id == sorted by destination rank)
        if Tl > 0:
            offs, tok, scale, pos = ops.moe_align(idx, w, E, False)
            send = ops.gather_rows(hl, tok)
            counts = (offs[1:E + 1] - offs[:E]).to(torch.int32)
        else:
            offs = tok = scale = pos = None
            send = h.new_zeros((0, H))
            counts = torch.zeros(E, dtype=torch.int32, device=h.device)
        # 2. who sends how many rows to which expert: [ep, E] on every rank (one small all-gather, then ONE host read)
        all_counts = (yield ("all_gather", counts)).view(ep, E).cpu()
        mine = all_counts[:, self.rank * El:(self.rank + 1) * El]            # [src, local expert]
        in_splits = all_counts[self.rank].view(ep, El).sum(1).tolist()       # rows I send to each rank
        out_splits = mine.sum(1).tolist()                                    # rows I receive from each rank
        recv = yield ("all_to_all", send, in_splits, out_splits)
        # 3. my experts over the received rows.  They arrive grouped by source, each block sorted by local expert:
        #    row index of (src, e) blocks -> gather list in expert order
        n_recv = int(sum(out_splits))
        if n_recv > 0:
            src_base = torch.tensor([0] + out_splits[:-1]).cumsum(0)          # first row of each source block
            within = torch.cat([torch.zeros(ep, 1, dtype=torch.long), mine.long().cumsum(1)[:, :-1]], dim=1)
            order = [torch.arange(int(mine[s, e])) + int(src_base[s]) + int(within[s, e])
                     for e in range(El) for s in range(ep)]
            gidx = torch.cat(order).to(torch.int32).to(h.device)
            loffs = torch.cat([torch.zeros(1, dtype=torch.long), mine.long().sum(0).cumsum(0)]).to(torch.int32).to(h.device)
            y = self._experts(recv, loffs, gidx, n_recv, None)                # unscaled: the weight is applied at home
            back = ops.scatter_rows(y, gidx, n_recv)                          # expert order -> arrival order
        else:
            back = h.new_zeros((0, H))
        ret = yield ("all_to_all", back, out_splits, in_splits)
        # 4. routing weights, ascending-expert combine, shared expert; then everyone gets every slice
        if Tl > 0:
            out_l = ops.moe_combine_scaled(ret, pos, scale, self.shared(hl), Tl, k)
        else:
            out_l = h.new_zeros((0, H))
        sizes = [token_slice(T, ep, r)[1] - token_slice(T, ep, r)[0] for r in range(ep)]
        out = yield ("all_gather_rows", out_l, sizes)
        return out


# ------------------------------------------------------------------------------------------------------------------
def _staged(group) -> bool:
    """gloo has no device all-to-all: the one-GPU rehearsal of the multi-rank control flow stages through the host."""
    return dist.get_backend(group) == "gloo"


def run(gen, group=None):
    """Drive one rank's generator with torch.distributed collectives (RCCL; gloo in CPU tests and rehearsals)."""
    host = _staged(group)
    try:
        req = next(gen)
        while True:
            op = req[0]
            dev = req[1].device
            stage = (lambda x: x.cpu()) if host and dev.type != "cpu" else (lambda x: x)
            if op == "all_gather":
                t = stage(req[1].contiguous())
                outl = torch.empty(dist.get_world_size(group) * t.numel(), dtype=t.dtype, device=t.device)
                dist.all_gather_into_tensor(outl, t, group=group)
                res = outl.to(dev)
            elif op == "all_to_all":
                _, t, in_splits, out_splits = req
                t = stage(t.contiguous())
                res = torch.empty((int(sum(out_splits)), t.shape[1]), dtype=t.dtype, device=t.device)
                dist.all_to_all_single(res, t, out_splits, in_splits, group=group)
                res = res.to(dev)
            elif op == "all_to_all_fixed":  # equal, capacity-sized slots: static shapes, no split lists
                t = stage(req[1].contiguous())
                res = torch.empty_like(t)
                dist.all_to_all_single(res, t, group=group)
                res = res.to(dev)
            elif op == "all_gather_rows":
                _, t, sizes = req
                per, world = max(sizes), len(sizes)  # slices are equal except the last: pad to one fixed-size gather
                padded = t if t.shape[0] == per else torch.cat([t, t.new_zeros((per - t.shape[0], t.shape[1]))])
                padded = stage(padded.contiguous())
                buf = torch.empty((world * per, t.shape[1]), dtype=t.dtype, device=padded.device)
                dist.all_gather_into_tensor(buf, padded, group=group)
                res = torch.cat([buf[r * per:r * per + n] for r, n in enumerate(sizes)], dim=0).to(dev)
            else:  # pragma: no cover
                raise RuntimeError(f"unknown collective {op}")
            req = gen.send(res)
    except StopIteration as e:
        return e.value


def run_emulated(gens: list):
    """Drive the generators of all logical ranks in lockstep inside ONE process (same collectives, done by copies)."""
    n = len(gens)
    reqs = [next(g) for g in gens]
    results = [None] * n
    done = 0
    while done < n:
        op = reqs[0][0]
        assert all(r[0] == op for r in reqs), "ranks diverged"
        if op == "all_gather":
            cat = torch.cat([r[1].reshape(-1) for r in reqs])
            res = [cat.clone() for _ in range(n)]
        elif op == "all_to_all":
            res = []
            for dst in range(n):
                pieces = []
                for src in range(n):
                    t, in_splits = reqs[src][1], reqs[src][2]
                    start = sum(in_splits[:dst])
                    pieces.append(t[start:start + in_splits[dst]])
                    assert in_splits[dst] == reqs[dst][3][src], "split bookkeeping disagrees between ranks"
                res.append(torch.cat(pieces, dim=0))
        elif op == "all_to_all_fixed":
            cap = reqs[0][2]
            res = [torch.cat([reqs[src][1][dst * cap:(dst + 1) * cap] for src in range(n)], dim=0) for dst in range(n)]
        elif op == "all_gather_rows":
            cat = torch.cat([r[1] for r in reqs], dim=0)
            res = [cat.clone() for _ in range(n)]
        else:  # pragma: no cover
            raise RuntimeError(f"unknown collective {op}")
        nxt = []
        for i, g in enumerate(gens):
            try:
                nxt.append(g.send(res[i]))
            except StopIteration as e:
                results[i] = e.value
                done += 1
                nxt.append(None)
        reqs = nxt
        if done and done < n:
            raise RuntimeError("ranks finished at different collectives")
    return results
