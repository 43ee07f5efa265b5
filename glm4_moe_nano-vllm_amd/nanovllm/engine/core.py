"""Host-side request bookkeeping for the runner: sequences, the paged-KV block pool with
prefix caching, and the prefill-first scheduler.

Behavioural contract = the reference's engine/sequence.py, engine/block_manager.py and
engine/scheduler.py (same public names and semantics: 256-token blocks, chained xxh64 block
hashes, prefill before decode, newest-first preemption).  The implementation is organised
around flat arrays instead of per-block objects; none of this is on the device hot path."""
from __future__ import annotations

import enum
import itertools
from collections import deque

import numpy as np
import xxhash

from nanovllm.sampling_params import SamplingParams


class SequenceStatus(enum.Enum):
    WAITING = enum.auto()
    RUNNING = enum.auto()
    FINISHED = enum.auto()


class Sequence:
    """One request: prompt + generated tokens + its physical block table."""
    block_size = 256
    counter = itertools.count()

    def __init__(self, token_ids: list[int], sampling_params: SamplingParams = SamplingParams()):
        self.seq_id = next(Sequence.counter)
        self.status = SequenceStatus.WAITING
        self.token_ids = list(token_ids)
        self.num_prompt_tokens = self.num_tokens = len(self.token_ids)
        self.last_token = self.token_ids[-1]
        self.num_cached_tokens = 0
        self.block_table: list[int] = []
        self.temperature = sampling_params.temperature
        self.max_tokens = sampling_params.max_tokens
        self.ignore_eos = sampling_params.ignore_eos

    # -- sizes ----------------------------------------------------------------------------
    def __len__(self) -> int:
        return self.num_tokens

    def __getitem__(self, key):
        return self.token_ids[key]

    @property
    def is_finished(self) -> bool:
        return self.status is SequenceStatus.FINISHED

    @property
    def num_completion_tokens(self) -> int:
        return self.num_tokens - self.num_prompt_tokens

    @property
    def prompt_token_ids(self) -> list[int]:
        return self.token_ids[: self.num_prompt_tokens]

    @property
    def completion_token_ids(self) -> list[int]:
        return self.token_ids[self.num_prompt_tokens:]

    @property
    def num_blocks(self) -> int:
        return -(-self.num_tokens // self.block_size)

    @property
    def num_cached_blocks(self) -> int:
        return self.num_cached_tokens // self.block_size

    @property
    def last_block_num_tokens(self) -> int:
        return self.num_tokens - (self.num_blocks - 1) * self.block_size

    def block(self, i: int) -> list[int]:
        if not 0 <= i < self.num_blocks:
            raise IndexError(i)
        return self.token_ids[i * self.block_size: (i + 1) * self.block_size]

    def append_token(self, token_id: int) -> None:
        self.token_ids.append(token_id)
        self.last_token = token_id
        self.num_tokens += 1

    # -- compact pickling for the rank-0 → worker control messages ------------------------------
    def __getstate__(self):
        body = self.token_ids if self.num_completion_tokens == 0 else self.last_token
        return {"n": self.num_tokens, "p": self.num_prompt_tokens, "c": self.num_cached_tokens,
                "bt": self.block_table, "t": self.temperature, "body": body}

    def __setstate__(self, s):
        self.num_tokens, self.num_prompt_tokens, self.num_cached_tokens = s["n"], s["p"], s["c"]
        self.block_table, self.temperature = s["bt"], s["t"]
        if isinstance(s["body"], list):
            self.token_ids = s["body"]
            self.last_token = self.token_ids[-1]
        else:
            self.last_token = s["body"]


class BlockManager:
    """Pool of `num_blocks` KV blocks.  Full blocks are content-addressed by a chained xxh64 of
    their tokens so that a later request with the same prefix re-uses them (prefix cache)."""
    NO_HASH = -1

    def __init__(self, num_blocks: int, block_size: int):
        self.block_size = block_size
        self.ref = [0] * num_blocks                 # reference counts
        self.hash = [self.NO_HASH] * num_blocks     # content hash of a full block
        self.tokens: list[list[int]] = [[] for _ in range(num_blocks)]
        self.hash_to_block_id: dict[int, int] = {}
        self.free_block_ids: deque[int] = deque(range(num_blocks))
        self.used_block_ids: set[int] = set()

    @classmethod
    def compute_hash(cls, token_ids: list[int], prefix: int = -1) -> int:
        h = xxhash.xxh64()
        if prefix != -1:
            h.update(prefix.to_bytes(8, "little"))
        h.update(np.array(token_ids).tobytes())
        return h.intdigest()

    # -- pool ------------------------------------------------------------------------------
    def _claim(self, bid: int) -> int:
        assert self.ref[bid] == 0
        self.ref[bid], self.hash[bid], self.tokens[bid] = 1, self.NO_HASH, []
        self.free_block_ids.remove(bid)
        self.used_block_ids.add(bid)
        return bid

    def _drop(self, bid: int) -> None:
        assert self.ref[bid] == 0
        self.used_block_ids.discard(bid)
        self.free_block_ids.append(bid)

    def _publish(self, bid: int, h: int, token_ids: list[int]) -> None:
        self.hash[bid], self.tokens[bid] = h, token_ids
        self.hash_to_block_id[h] = bid

    # -- prefill ---------------------------------------------------------------------------
    def can_allocate(self, seq: Sequence) -> bool:
        return len(self.free_block_ids) >= seq.num_blocks

    def allocate(self, seq: Sequence) -> None:
        assert not seq.block_table
        chain, hit = self.NO_HASH, True
        for i in range(seq.num_blocks):
            toks = seq.block(i)
            chain = self.compute_hash(toks, chain) if len(toks) == self.block_size else self.NO_HASH
            bid = self.hash_to_block_id.get(chain, -1) if chain != self.NO_HASH else -1
            hit = hit and bid != -1 and self.tokens[bid] == toks
            if not hit:
                bid = self._claim(self.free_block_ids[0])
            else:
                seq.num_cached_tokens += self.block_size
                if bid in self.used_block_ids:
                    self.ref[bid] += 1
                else:
                    self._claim(bid)
            if chain != self.NO_HASH:
                self._publish(bid, chain, toks)
            seq.block_table.append(bid)

    def deallocate(self, seq: Sequence) -> None:
        for bid in reversed(seq.block_table):
            self.ref[bid] -= 1
            if self.ref[bid] == 0:
                self._drop(bid)
        seq.num_cached_tokens = 0
        seq.block_table.clear()

    # -- decode ----------------------------------------------------------------------------
    def can_append(self, seq: Sequence) -> bool:
        needs_new_block = len(seq) % self.block_size == 1
        return len(self.free_block_ids) >= int(needs_new_block)

    def may_append(self, seq: Sequence) -> None:
        table = seq.block_table
        fill = len(seq) % self.block_size
        if fill == 1:       # the token just appended opens a new block
            assert self.hash[table[-1]] != self.NO_HASH
            table.append(self._claim(self.free_block_ids[0]))
        elif fill == 0:     # the last block just became full: make it shareable
            assert self.hash[table[-1]] == self.NO_HASH
            toks = seq.block(seq.num_blocks - 1)
            prefix = self.hash[table[-2]] if len(table) > 1 else -1
            self._publish(table[-1], self.compute_hash(toks, prefix), toks)
        else:
            assert self.hash[table[-1]] == self.NO_HASH


class Scheduler:

    def __init__(self, config):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.eos = config.eos
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting: deque[Sequence] = deque()
        self.running: deque[Sequence] = deque()

    def is_finished(self) -> bool:
        return not (self.waiting or self.running)

    def add(self, seq: Sequence) -> None:
        self.waiting.append(seq)

    def _admit_prefill(self) -> list[Sequence]:
        batch, budget = [], self.max_num_batched_tokens
        while self.waiting and len(batch) < self.max_num_seqs:
            seq = self.waiting[0]
            if len(seq) > budget or not self.block_manager.can_allocate(seq):
                break
            self.block_manager.allocate(seq)
            budget -= len(seq) - seq.num_cached_tokens
            seq.status = SequenceStatus.RUNNING
            self.running.append(self.waiting.popleft())
            batch.append(seq)
        return batch

    def _admit_decode(self) -> list[Sequence]:
        batch: list[Sequence] = []
        while self.running and len(batch) < self.max_num_seqs:
            seq = self.running.popleft()
            evicted_self = False
            while not self.block_manager.can_append(seq):
                if self.running:
                    self.preempt(self.running.pop())   # newest request gives up its blocks
                else:
                    self.preempt(seq)
                    evicted_self = True
                    break
            if not evicted_self:
                self.block_manager.may_append(seq)
                batch.append(seq)
        assert batch
        self.running.extendleft(reversed(batch))
        return batch

    def schedule(self) -> tuple[list[Sequence], bool]:
        batch = self._admit_prefill()
        if batch:
            return batch, True
        return self._admit_decode(), False

    def preempt(self, seq: Sequence) -> None:
        seq.status = SequenceStatus.WAITING
        self.block_manager.deallocate(seq)
        self.waiting.appendleft(seq)

    def postprocess(self, seqs: list[Sequence], token_ids: list[int]) -> None:
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok)
            hit_eos = tok == self.eos and not seq.ignore_eos
            if hit_eos or seq.num_completion_tokens == seq.max_tokens:
                seq.status = SequenceStatus.FINISHED
                self.block_manager.deallocate(seq)
                self.running.remove(seq)
