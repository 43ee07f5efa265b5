"""Host-side scheduling core (engine/core.py): block pool, prefix cache, scheduler policy."""
import types

from nanovllm.engine.block_manager import BlockManager
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import Sequence, SequenceStatus
from nanovllm.sampling_params import SamplingParams


def cfg(**kw):
    base = dict(max_num_seqs=4, max_num_batched_tokens=2048, eos=-1, num_kvcache_blocks=8, kvcache_block_size=256)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_sequence_block_views_and_pickle():
    import pickle
    s = Sequence(list(range(600)), SamplingParams(temperature=0.0, max_tokens=3))
    assert s.num_blocks == 3 and s.last_block_num_tokens == 600 - 512 and len(s.block(2)) == 88
    s2 = pickle.loads(pickle.dumps(s))
    assert s2.token_ids == s.token_ids and s2.last_token == 599 and s2.temperature == 0.0
    s.append_token(7)
    s3 = pickle.loads(pickle.dumps(s))  # after the first completion only the last token travels
    assert s3.last_token == 7 and s3.num_tokens == 601 and not hasattr(s3, "token_ids")


def test_prefix_cache_shares_full_blocks():
    bm = BlockManager(8, 256)
    a = Sequence(list(range(600)))
    bm.allocate(a)
    assert a.num_cached_tokens == 0 and len(a.block_table) == 3
    b = Sequence(list(range(512)) + [9999] * 50)
    bm.allocate(b)  # first two full blocks identical → shared
    assert b.num_cached_tokens == 512 and b.block_table[:2] == a.block_table[:2] and b.block_table[2] != a.block_table[2]
    assert bm.ref[a.block_table[0]] == 2
    bm.deallocate(a)
    assert bm.ref[b.block_table[0]] == 1 and len(bm.free_block_ids) == 8 - 3
    c = Sequence(list(range(256)) + [5] * 10)
    bm.allocate(c)  # block 0 content still cached
    assert c.num_cached_tokens == 256 and c.block_table[0] == b.block_table[0]


def test_may_append_opens_and_seals_blocks():
    bm = BlockManager(4, 256)
    s = Sequence(list(range(255)))
    bm.allocate(s)
    assert len(s.block_table) == 1 and bm.hash[s.block_table[0]] == BlockManager.NO_HASH
    s.append_token(1)            # 256 tokens: block becomes full → sealed (hashed)
    bm.may_append(s)
    assert bm.hash[s.block_table[0]] != BlockManager.NO_HASH and len(s.block_table) == 1
    s.append_token(2)            # 257: needs a new block
    assert bm.can_append(s)
    bm.may_append(s)
    assert len(s.block_table) == 2


def test_scheduler_prefill_first_then_decode_and_finish():
    sch = Scheduler(cfg())
    sp = SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True)
    for n in (10, 300):
        sch.add(Sequence(list(range(n)), sp))
    seqs, is_prefill = sch.schedule()
    assert is_prefill and [len(s) for s in seqs] == [10, 300]
    sch.postprocess(seqs, [1, 2])
    seqs, is_prefill = sch.schedule()
    assert not is_prefill and len(seqs) == 2
    sch.postprocess(seqs, [3, 4])
    assert sch.is_finished() and all(s.status is SequenceStatus.FINISHED for s in seqs)
    assert len(sch.block_manager.free_block_ids) == 8


def test_scheduler_preempts_newest_when_out_of_blocks():
    sch = Scheduler(cfg(num_kvcache_blocks=2))
    sp = SamplingParams(temperature=0.0, max_tokens=8, ignore_eos=True)
    a, b = Sequence(list(range(256)), sp), Sequence(list(range(1000, 1256)), sp)
    sch.add(a); sch.add(b)
    seqs, is_prefill = sch.schedule()
    assert is_prefill and len(seqs) == 2          # both blocks in use, both full
    sch.postprocess(seqs, [1, 1])                 # 257 tokens each → each needs a new block
    seqs, is_prefill = sch.schedule()
    assert not is_prefill and seqs == [a]         # b was preempted to make room
    assert b.status is SequenceStatus.WAITING and not b.block_table
