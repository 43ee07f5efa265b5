"""User-facing engine — reference: engine/llm_engine.py:15-93 (same constructor keywords,
`add_request`, `step`, `generate` and output records)."""
from __future__ import annotations

import atexit
import os
import socket
from dataclasses import fields
from time import perf_counter

import torch.multiprocessing as mp

from nanovllm.config import Config
from nanovllm.engine.model_runner import ModelRunner
from nanovllm.engine.scheduler import Scheduler
from nanovllm.engine.sequence import Sequence
from nanovllm.sampling_params import SamplingParams


def _worker(config: Config, rank: int, port: int):
    ModelRunner(config, rank, port).loop()


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class LLMEngine:

    def __init__(self, model, **kwargs):
        names = {f.name for f in fields(Config)}
        config = Config(model, **{k: v for k, v in kwargs.items() if k in names})
        self.config = config
        self.ps = []
        port = _free_port()
        if config.tensor_parallel_size > 1:
            ctx = mp.get_context("spawn")
            for r in range(1, config.tensor_parallel_size):
                p = ctx.Process(target=_worker, args=(config, r, port))
                p.start()
                self.ps.append(p)
        self.model_runner = ModelRunner(config, 0, port)
        self.tokenizer = self._load_tokenizer(config.model)
        if self.tokenizer is not None and self.tokenizer.eos_token_id is not None:
            config.eos = self.tokenizer.eos_token_id
        self.scheduler = Scheduler(config)
        self._closed = False
        atexit.register(self.exit)

    @staticmethod
    def _load_tokenizer(path):
        if not any(os.path.exists(os.path.join(path, f)) for f in ("tokenizer.json", "tokenizer_config.json",
                                                                    "tokenizer.model")):
            return None  # token-id-only use (synthetic checkpoints)
        from transformers import AutoTokenizer
        return AutoTokenizer.from_pretrained(path, use_fast=True)

    def exit(self):
        if self._closed:
            return
        self._closed = True
        self.model_runner.call("exit")
        del self.model_runner
        for p in self.ps:
            p.join()

    def add_request(self, prompt: str | list[int], sampling_params: SamplingParams):
        if isinstance(prompt, str):
            if self.tokenizer is None:
                raise ValueError("this model directory has no tokenizer: pass token id lists")
            prompt = self.tokenizer.encode(prompt)
        self.scheduler.add(Sequence(prompt, sampling_params))

    def step(self):
        seqs, is_prefill = self.scheduler.schedule()
        token_ids = self.model_runner.call("run", seqs, is_prefill)
        self.scheduler.postprocess(seqs, token_ids)
        outputs = [(s.seq_id, s.completion_token_ids) for s in seqs if s.is_finished]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return outputs, num_tokens

    def is_finished(self):
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params, use_tqdm: bool = True):
        pbar = None
        if use_tqdm:
            from tqdm.auto import tqdm
            pbar = tqdm(total=len(prompts), desc="Generating", dynamic_ncols=True)
        if not isinstance(sampling_params, list):
            sampling_params = [sampling_params] * len(prompts)
        for prompt, sp in zip(prompts, sampling_params):
            self.add_request(prompt, sp)
        done = {}
        prefill_tps = decode_tps = 0.0
        while not self.is_finished():
            t0 = perf_counter()
            finished, num_tokens = self.step()
            dt = perf_counter() - t0
            if num_tokens > 0:
                prefill_tps = num_tokens / dt
            else:
                decode_tps = -num_tokens / dt
            if pbar is not None:
                pbar.set_postfix({"Prefill": f"{int(prefill_tps)}tok/s", "Decode": f"{int(decode_tps)}tok/s"})
                pbar.update(len(finished))
            done.update(finished)
        if pbar is not None:
            pbar.close()
        self.last_throughput = {"prefill_tok_s": prefill_tps, "decode_tok_s": decode_tps}
        decode = (lambda ids: self.tokenizer.decode(ids)) if self.tokenizer is not None else (lambda ids: "")
        return [{"text": decode(done[k]), "token_ids": done[k]} for k in sorted(done)]
