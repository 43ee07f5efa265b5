"""End-to-end `nanovllm.LLM.generate()` on a tiny synthetic GLM-4-MoE checkpoint: prefill →
paged decode through the HIP path, eager and HIP-graph modes, checked against the CPU oracle's
greedy decode.  A generated token may differ from the oracle's only where the oracle's own top-2
logit margin is below bf16 noise (0.06); comparison of that sequence stops there."""
import pytest
import torch

from util import (oracle_layers_from_checkpoint, oracle_next_token_logits, tiny_model_config,
                  write_tiny_checkpoint)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def checkpoint(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("tiny_glm4moe"))
    cfg = tiny_model_config()
    tensors = write_tiny_checkpoint(path, cfg, seed=3)
    return path, cfg, tensors


def _generate(path, eager, prompts, max_tokens):
    from nanovllm import LLM, SamplingParams
    llm = LLM(path, enforce_eager=eager, max_model_len=1024, max_num_batched_tokens=2048, max_num_seqs=8,
              num_kvcache_blocks=32)
    try:
        out = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=max_tokens, ignore_eos=True),
                           use_tqdm=False)
    finally:
        llm.exit()
    torch.cuda.empty_cache()
    return [o["token_ids"] for o in out]


def _compare_with_oracle(t, cfg, layers, prompt, got) -> int:
    """Greedy tokens vs the CPU oracle's; a token may differ only where the oracle's own top-2 margin is below bf16
    noise, and the comparison of that sequence stops there.  Returns the number of tokens compared equal."""
    n, seq = 0, list(prompt)
    for tok in got:
        logits = oracle_next_token_logits(t, cfg, layers, seq).float()
        top2 = torch.topk(logits, 2)
        if int(top2.indices[0]) != tok:
            margin = float(top2.values[0] - logits[tok])
            assert margin < 0.06, f"token {tok} vs oracle {int(top2.indices[0])}, margin {margin:.3f}"
            break
        n += 1
        seq.append(tok)
    return n


def test_generate_matches_oracle_greedy(checkpoint):
    path, cfg, t = checkpoint
    g = torch.Generator().manual_seed(11)
    prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in (5, 300, 17, 256)]
    max_tokens = 5
    eager = _generate(path, True, prompts, max_tokens)
    graph = _generate(path, False, prompts, max_tokens)
    assert eager == graph, "HIP-graph decode must reproduce the eager tokens exactly (deterministic kernels)"
    layers = oracle_layers_from_checkpoint(t, cfg)
    n_checked = 0
    for prompt, got in zip(prompts, eager):
        assert len(got) == max_tokens
        seq = list(prompt)
        for tok in got:
            logits = oracle_next_token_logits(t, cfg, layers, seq).float()
            top2 = torch.topk(logits, 2)
            if int(top2.indices[0]) != tok:
                margin = float(top2.values[0] - logits[tok])
                assert margin < 0.06, f"token {tok} vs oracle {int(top2.indices[0])}, margin {margin:.3f}"
                break
            n_checked += 1
            seq.append(tok)
    assert n_checked >= 12, f"only {n_checked} tokens could be compared"


def test_prefix_cache_prefill_path(checkpoint):
    """Second request shares a full 256-token block with the first → prefill with block_tables
    (paged K/V) must give the same continuation as without sharing."""
    path, cfg, t = checkpoint
    g = torch.Generator().manual_seed(12)
    common = torch.randint(0, cfg["vocab_size"], (256,), generator=g).tolist()
    tail = torch.randint(0, cfg["vocab_size"], (40,), generator=g).tolist()
    from nanovllm import LLM, SamplingParams
    sp = SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True)
    llm = LLM(path, enforce_eager=True, max_model_len=1024, max_num_batched_tokens=2048, max_num_seqs=8,
              num_kvcache_blocks=32)
    try:
        first = llm.generate([common + tail], sp, use_tqdm=False)[0]["token_ids"]
        # the 256-token block is cached now; same prompt again goes through the prefix-cache branch
        again = llm.generate([common + tail], sp, use_tqdm=False)[0]["token_ids"]
    finally:
        llm.exit()
    assert first == again
    # ... and both are what the reference algorithm generates (the second run attended over PAGED prefix K/V)
    layers = oracle_layers_from_checkpoint(t, cfg)
    assert _compare_with_oracle(t, cfg, layers, common + tail, again) >= 2
