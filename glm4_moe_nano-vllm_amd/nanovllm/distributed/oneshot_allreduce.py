"""One-shot all-reduce over directly mapped peer buffers (csrc/oneshot_allreduce.hip) — OPT-IN (GLM_ONESHOT_ALLREDUCE=1).

RCCL (`dist.all_reduce`) is the default and the only path measured on more than one GPU.  This module maps every rank's
receive buffer and flag array into every peer (PyTorch's CUDA-IPC storage sharing, dmabuf on this ROCm), after which an
all-reduce of a [T, H] bf16 tensor is ONE kernel per rank: push the rows into every peer's receive slot, signal the
peers, wait for their signals, sum the contributions out of local memory in rank order.  The call counter (epoch) lives
in DEVICE memory and is advanced by the kernel, so captured calls stay correct under HIP-graph replay.  Anything that
goes wrong while setting it up leaves the TP group on RCCL; a wait that expires at run time poisons the output with NaNs
and sets a status word that `check()` turns into an exception at the engine's next synchronisation point.

Tested: world = 1; two "ranks" on two streams of one device, one of them delayed (tests/test_hip_kernels.py); two
PROCESSES sharing one device over CUDA-IPC, eager and under graph replay with an odd number of calls per graph
(tools/oneshot_allreduce_2proc.py on the GPU box).  Not yet run across xGMI."""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

from nanovllm import hip_ops

N_FLAGS = 8


class OneShotAllReduce:

    def __init__(self, n_max: int, device, rank: int, world: int):
        assert n_max % 8 == 0 and 1 <= world <= N_FLAGS
        self.n_max, self.device, self.rank, self.world = n_max, device, rank, world
        # receive buffer [2 slots][world senders][n_max]; flags[q] = last epoch complete from rank q; state = epoch,
        # arrive, done, status (all advanced by the kernel)
        self.buf = torch.zeros(2 * world * n_max, dtype=torch.bfloat16, device=device)
        self.flags = torch.zeros(N_FLAGS, dtype=torch.int32, device=device)
        self.state = torch.zeros(4, dtype=torch.int32, device=device)
        self.peer_bufs = [self.buf] * world      # replaced by connect()
        self.peer_flags = [self.flags] * world
        self._keep = []

    # ---- wiring -------------------------------------------------------------------------------------------------
    def share_handles(self):
        """What a peer needs to map this rank's buffer and flags (picklable)."""
        return (self.buf.untyped_storage()._share_cuda_(), self.flags.untyped_storage()._share_cuda_())

    @staticmethod
    def _open(handle, dtype, numel, device):
        st = torch.UntypedStorage._new_shared_cuda(*handle)
        return torch.empty(0, dtype=dtype, device=device).set_(st, 0, (numel,))

    def connect(self, handles: list):
        """handles[r] = rank r's share_handles() (own entry ignored)."""
        bufs, flags = [], []
        for r, h in enumerate(handles):
            if r == self.rank:
                bufs.append(self.buf)
                flags.append(self.flags)
            else:
                bufs.append(self._open(h[0], torch.bfloat16, 2 * self.world * self.n_max, self.device))
                flags.append(self._open(h[1], torch.int32, N_FLAGS, self.device))
        self.peer_bufs, self.peer_flags = bufs, flags

    @classmethod
    def for_group(cls, n_max: int, device, group=None):
        """Collective constructor: exchanges the IPC handles through torch.distributed.  Failure-symmetric: every rank
        takes part in the same two object all-gathers whatever happens locally, and all raise together."""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        ar, mine = None, None
        try:
            ar = cls(n_max, device, rank, world)
            mine = ar.share_handles()
        except Exception as e:  # noqa: BLE001
            mine = None
            err = repr(e)
        handles = [None] * world
        dist.all_gather_object(handles, mine, group=group)
        ok = 0
        if all(h is not None for h in handles):
            try:
                ar.connect(handles)
                ok = 1
            except Exception as e:  # noqa: BLE001
                err = repr(e)
        oks = [None] * world
        dist.all_gather_object(oks, ok, group=group)
        if not all(oks):
            raise RuntimeError(f"peer mapping failed on ranks {[r for r, o in enumerate(oks) if not o]}"
                               + (f" (here: {err})" if not ok and "err" in dir() else ""))
        return ar

    # ---- the collective -----------------------------------------------------------------------------------------
    def all_reduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks (bf16, numel <= n_max, multiple of 8, contiguous).  Graph-capturable: nothing
        about the call is a host-side counter."""
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() % 8 == 0 and t.numel() <= self.n_max
        P = ctypes.c_void_p * self.world
        pb = P(*[b.data_ptr() for b in self.peer_bufs])
        pf = P(*[f.data_ptr() for f in self.peer_flags])
        rc = hip_ops.load_library().glm_allreduce_oneshot(
            t.data_ptr(), t.data_ptr(), pb, pf, self.rank, self.world, t.numel(), self.n_max, self.state.data_ptr(),
            torch.cuda.current_stream().cuda_stream)
        hip_ops._check(rc, "glm_allreduce_oneshot")
        return t

    def status(self) -> int:
        """0 = every wait completed; k + 1 = some call gave up waiting for rank k (synchronises)."""
        return int(self.state[3].item())

    def epoch(self) -> int:
        """Completed calls so far (synchronises)."""
        return int(self.state[0].item())

    def check(self) -> None:
        """Raise if a call gave up waiting (its output was poisoned with NaNs).  Call at a point that synchronises anyway."""
        st = self.status()
        if st:
            raise RuntimeError(f"one-shot all-reduce: rank {self.rank} gave up waiting for rank {st - 1}; "
                               "the outputs of that call are NaN")


def enabled() -> bool:
    return os.environ.get("GLM_ONESHOT_ALLREDUCE", "0") == "1"
