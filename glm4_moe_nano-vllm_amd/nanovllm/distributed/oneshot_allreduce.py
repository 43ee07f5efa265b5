"""One-shot all-reduce over directly mapped peer buffers (csrc/oneshot_allreduce.hip) — OPT-IN (GLM_ONESHOT_ALLREDUCE=1).

RCCL (`dist.all_reduce`) is the default and the only path measured on more than one GPU.  This module maps every rank's
comm buffer and flag array into every peer (PyTorch's CUDA-IPC storage sharing, dmabuf on this ROCm), after which an
all-reduce of a [T, H] bf16 tensor is ONE kernel per rank: publish, signal the 7 peers, wait for their signals, sum all
buffers in rank order.  Anything that goes wrong while setting it up leaves the TP group on RCCL.

Tested: world = 1; two "ranks" on two streams of one device (tests/test_hip_kernels.py); two PROCESSES sharing one
device over CUDA-IPC (tools/oneshot_allreduce_2proc.py on the GPU box).  Not yet run across xGMI."""
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
        self.buf = torch.zeros(2 * n_max, dtype=torch.bfloat16, device=device)
        self.flags = torch.zeros(N_FLAGS, dtype=torch.int32, device=device)
        self.local = torch.zeros(4, dtype=torch.int32, device=device)   # [0] arrival counter, [1] status
        self.peer_bufs = [self.buf] * world      # replaced by connect()
        self.peer_flags = [self.flags] * world
        self.epoch = 0
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
                bufs.append(self._open(h[0], torch.bfloat16, 2 * self.n_max, self.device))
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
        """In-place sum over the ranks (bf16, numel <= n_max, multiple of 8, contiguous)."""
        assert t.dtype == torch.bfloat16 and t.is_contiguous() and t.numel() % 8 == 0 and t.numel() <= self.n_max
        self.epoch += 1
        P = ctypes.c_void_p * self.world
        pb = P(*[b.data_ptr() for b in self.peer_bufs])
        pf = P(*[f.data_ptr() for f in self.peer_flags])
        rc = hip_ops.load_library().glm_allreduce_oneshot(
            t.data_ptr(), t.data_ptr(), pb, pf, self.rank, self.world, t.numel(), self.n_max, self.epoch,
            self.local.data_ptr(), self.local[1:].data_ptr(), torch.cuda.current_stream().cuda_stream)
        hip_ops._check(rc, "glm_allreduce_oneshot")
        return t

    def status(self) -> int:
        """0 = every wait completed; k + 1 = some call gave up waiting for rank k (synchronises)."""
        return int(self.local[1].item())


def enabled() -> bool:
    return os.environ.get("GLM_ONESHOT_ALLREDUCE", "0") == "1"
