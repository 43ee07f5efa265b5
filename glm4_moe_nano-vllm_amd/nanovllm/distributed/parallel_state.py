"""Tensor-parallel group.  The reference ships a stub that always answers world_size=1
(distributed/parallel_state.py:1-9), so its Column/Row/QKV linears never shard.  Here the same
accessor returns the REAL group: one process per GPU, torch.distributed ("nccl" = RCCL over
xGMI on ROCm; "gloo" in CPU tests), TP group = the whole world (single node, ≤ 8 ranks,
config.py:23)."""
from __future__ import annotations

import torch
import torch.distributed as dist


class TPGroup:
    """Minimal coordinator: world_size / rank plus the one collective the layer needs
    (all-reduce at RowParallel sites, layers/linear.py:153-154)."""

    def __init__(self, group=None):
        self._group = group

    @property
    def process_group(self):
        """The torch.distributed group behind this coordinator (None = the world): collectives other than the
        all-reduce (the expert-parallel all-to-all) run on the same group."""
        return self._group

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self._group) if dist.is_available() and dist.is_initialized() else 1

    @property
    def rank(self) -> int:
        return dist.get_rank(self._group) if dist.is_available() and dist.is_initialized() else 0

    _oneshot = None          # OneShotAllReduce once set up; False = tried and unavailable
    ONESHOT_MAX_ELEMS = 512 * 8192

    def _oneshot_for(self, t: torch.Tensor):
        """Opt-in (GLM_ONESHOT_ALLREDUCE=1) one-hop all-reduce over mapped peer buffers for the small [T, H] tensors of
        the RowParallel sites; RCCL otherwise.  Set up collectively at the first eligible call; any failure keeps RCCL."""
        if self._oneshot is False:
            return None
        from nanovllm.distributed import oneshot_allreduce as osa
        if not osa.enabled() or not t.is_cuda or t.dtype != torch.bfloat16 or not t.is_contiguous() or \
                t.numel() % 8 or t.numel() > self.ONESHOT_MAX_ELEMS or self.world_size > osa.N_FLAGS:
            return None
        if self._oneshot is None:
            try:
                self._oneshot = osa.OneShotAllReduce.for_group(self.ONESHOT_MAX_ELEMS, t.device, self._group)
            except Exception as e:  # noqa: BLE001 - stay on RCCL
                print(f"[nanovllm] one-shot all-reduce unavailable ({e!r}); using RCCL")
                self._oneshot = False
                return None
        return self._oneshot

    def check(self) -> None:
        """Surface a failed one-shot all-reduce (bounded wait expired: outputs poisoned) as an exception.  Cheap no-op
        on RCCL; the engine calls it where it synchronises anyway (after reading the sampled tokens)."""
        if self._oneshot:
            self._oneshot.check()

    # live counters of the calls this coordinator issued (host side; a captured step counts once, at capture): what
    # bench.py's `collective` record is filled from, so that an N > 1 line shows what actually ran
    calls = 0
    calls_oneshot = 0
    bytes_total = 0
    bytes_max = 0

    def reset_counters(self) -> None:
        self.calls = self.calls_oneshot = self.bytes_total = self.bytes_max = 0

    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        if self.world_size > 1:
            nb = t.numel() * t.element_size()
            self.calls += 1
            self.bytes_total += nb
            self.bytes_max = max(self.bytes_max, nb)
            fast = self._oneshot_for(t)
            if fast is not None:
                self.calls_oneshot += 1
                return fast.all_reduce(t)
            dist.all_reduce(t, group=self._group)
        return t


_TP_GROUP = TPGroup()


def get_tp_group() -> TPGroup:
    return _TP_GROUP


def set_tp_group(group) -> None:
    """Bind the TP coordinator to a specific process group (default: the world)."""
    global _TP_GROUP
    _TP_GROUP = TPGroup(group)


class _SingleRank:
    """What the reference's stub always answers (parallel_state.py:1-9): used while building modules that must NOT be
    tensor-parallel although the process group is (the experts of an expert-parallel MoE block)."""
    world_size, rank, process_group = 1, 0, None

    def all_reduce(self, t):
        return t

    def check(self):
        pass


class unsharded:
    """Context manager: modules constructed inside see a single-rank TP group (their linears keep full shapes)."""

    def __enter__(self):
        global _TP_GROUP
        self._saved = _TP_GROUP
        _TP_GROUP = _SingleRank()

    def __exit__(self, *exc):
        global _TP_GROUP
        _TP_GROUP = self._saved
        return False


def tensor_model_parallel_all_reduce(t: torch.Tensor) -> torch.Tensor:
    return get_tp_group().all_reduce(t)
