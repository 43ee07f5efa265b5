"""One-shot all-reduce over directly mapped peer buffers (csrc/oneshot_allreduce.hip) — OPT-IN (GLM_ONESHOT_ALLREDUCE=1).

RCCL (`dist.all_reduce`) is the default and the only path measured on more than one GPU.  This module maps every rank's
receive buffer and flag array — UNCACHED device memory from the library's glm_comm_* helpers (r04; r03 used ordinary torch
tensors, whose lines the owner's L2 may keep across a running kernel) — into every peer over hipIpc, after which an
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
COMM_UNCACHED, COMM_FINEGRAINED = 0, 1
IPC_HANDLE_BYTES = 64


class CommBuffer:
    """Peer-written communication memory (include/glm4moe_hip.h, glm_comm_*): device memory allocated UNCACHED (or
    fine-grained with GLM_ONESHOT_COMM_KIND=finegrained) instead of as an ordinary, L2-cacheable torch tensor — a peer GPU
    writes it over xGMI while the owner polls and reads it inside a running kernel.  Zero-filled.  `handle()` is an
    opaque bytes object another process turns into a mapping of the same memory with `CommBuffer.open()`."""

    def __init__(self, nbytes: int, device, kind: int | None = None, _ptr: int | None = None, _owner: bool = True):
        self.nbytes, self.device, self._owner = int(nbytes), torch.device(device), _owner
        self._lib = hip_ops.load_library()
        if _ptr is not None:
            self.ptr = _ptr
            return
        if kind is None:
            kind = COMM_FINEGRAINED if os.environ.get("GLM_ONESHOT_COMM_KIND", "uncached") == "finegrained" else COMM_UNCACHED
        self.kind = kind
        p = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            hip_ops._check(self._lib.glm_comm_alloc(ctypes.byref(p), self.nbytes, kind), "glm_comm_alloc")
        self.ptr = p.value

    def data_ptr(self) -> int:
        return self.ptr

    def handle(self) -> bytes:
        h = ctypes.create_string_buffer(IPC_HANDLE_BYTES)
        hip_ops._check(self._lib.glm_comm_ipc_export(ctypes.c_void_p(self.ptr), h), "glm_comm_ipc_export")
        return h.raw

    @classmethod
    def open(cls, handle: bytes, nbytes: int, device) -> "CommBuffer":
        lib = hip_ops.load_library()
        p = ctypes.c_void_p()
        with torch.cuda.device(torch.device(device)):
            hip_ops._check(lib.glm_comm_ipc_open(ctypes.create_string_buffer(handle, IPC_HANDLE_BYTES), ctypes.byref(p)),
                           "glm_comm_ipc_open")
        return cls(nbytes, device, _ptr=p.value, _owner=False)

    def close(self) -> None:
        ptr, self.ptr = getattr(self, "ptr", None), None
        if not ptr:
            return
        try:
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()
                (self._lib.glm_comm_free if self._owner else self._lib.glm_comm_ipc_close)(ctypes.c_void_p(ptr))
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass

    def __del__(self):
        self.close()


class OneShotAllReduce:

    def __init__(self, n_max: int, device, rank: int, world: int):
        assert n_max % 8 == 0 and 1 <= world <= N_FLAGS
        self.n_max, self.device, self.rank, self.world = n_max, device, rank, world
        # receive buffer [2 slots][world senders][n_max] bf16 and flags[q] = last epoch complete from rank q: peer-written
        # => communication memory (uncached), not torch storage.  state = epoch, arrive, done, status: touched by this
        # rank's own kernel only => an ordinary tensor.
        self.buf = CommBuffer(2 * world * n_max * 2, device)
        self.flags = CommBuffer(N_FLAGS * 4, device)
        self.state = torch.zeros(4, dtype=torch.int32, device=device)
        self.peer_bufs = [self.buf] * world      # replaced by connect()
        self.peer_flags = [self.flags] * world

    # ---- wiring -------------------------------------------------------------------------------------------------
    def share_handles(self):
        """What a peer needs to map this rank's buffer and flags (picklable)."""
        return (self.buf.handle(), self.flags.handle())

    def connect(self, handles: list):
        """handles[r] = rank r's share_handles() (own entry ignored)."""
        bufs, flags = [], []
        for r, h in enumerate(handles):
            if r == self.rank:
                bufs.append(self.buf)
                flags.append(self.flags)
            else:
                bufs.append(CommBuffer.open(h[0], self.buf.nbytes, self.device))
                flags.append(CommBuffer.open(h[1], self.flags.nbytes, self.device))
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
        ar._group_made = True
        return ar

    def close(self, group=None) -> None:
        """Collective teardown (r05).  Order matters: EVERY rank first closes its mappings of the peers' blocks
        (glm_comm_ipc_close), the ranks meet, and only then does an owner hand its own blocks back (glm_comm_free parks
        them for the next glm_comm_alloc, which re-zeroes them: a peer that still had the block mapped could write flags or
        rows of an old epoch into the new owner's buffer).  Without a process group (single process, emulated ranks) the
        caller closes every rank's object before it allocates again.  Operating rule: a serving
        process allocates its communication blocks once at start-up and calls this at shutdown only."""
        torch.cuda.synchronize(self.device)
        for lst in (self.peer_bufs, self.peer_flags):
            for b in lst:
                if b is not None and not b._owner:
                    b.close()
        self.peer_bufs, self.peer_flags = [self.buf] * self.world, [self.flags] * self.world
        if getattr(self, "_group_made", False):   # built by for_group(): the peers are other processes
            dist.barrier(group=group)
        self.buf.close()
        self.flags.close()

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
    return hip_ops.opt("oneshot_allreduce") == 1
