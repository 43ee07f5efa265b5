"""Engine configuration — reference: config.py:6-26.  `hf_config` is normalised into a plain
namespace with the attribute names the layer code reads (transformers 5.x moved rope_theta into
`rope_parameters`; the reference was written against the 4.x names)."""
from __future__ import annotations

import json
import os
import types
from dataclasses import dataclass

import torch


def normalise_hf_config(cfg) -> types.SimpleNamespace:
    d = cfg if isinstance(cfg, dict) else (cfg.to_dict() if hasattr(cfg, "to_dict") else dict(vars(cfg)))
    d = dict(d)
    rp = d.get("rope_parameters") or {}
    if "rope_theta" not in d or d["rope_theta"] is None:
        d["rope_theta"] = rp.get("rope_theta", 10000.0)
    if "partial_rotary_factor" not in d or d["partial_rotary_factor"] is None:
        d["partial_rotary_factor"] = rp.get("partial_rotary_factor", 0.5)
    rs = d.get("rope_scaling")
    if isinstance(rs, dict) and rs.get("rope_type", rs.get("type", "default")) == "default":
        rs = None
    d["rope_scaling"] = rs
    d.setdefault("head_dim", d["hidden_size"] // d["num_attention_heads"])
    d.setdefault("attention_bias", False)
    d.setdefault("use_qk_norm", False)
    d.setdefault("n_group", 1)
    d.setdefault("topk_group", 1)
    d.setdefault("routed_scaling_factor", 1.0)
    d.setdefault("first_k_dense_replace", 0)
    d.setdefault("hidden_act", "silu")
    d.setdefault("tie_word_embeddings", False)
    dt = d.get("torch_dtype") or d.get("dtype") or "bfloat16"
    d["torch_dtype"] = getattr(torch, dt) if isinstance(dt, str) else dt
    return types.SimpleNamespace(**d)


@dataclass
class Config:
    model: str
    max_num_batched_tokens: int = 16384
    max_num_seqs: int = 512
    max_model_len: int = 4096
    gpu_memory_utilization: float = 0.9
    tensor_parallel_size: int = 1
    enforce_eager: bool = False
    hf_config: object | None = None
    eos: int = -1
    kvcache_block_size: int = 256
    num_kvcache_blocks: int = -1

    def __post_init__(self):
        assert os.path.isdir(self.model)
        assert self.kvcache_block_size % 256 == 0
        assert 1 <= self.tensor_parallel_size <= 8
        if self.hf_config is None:
            with open(os.path.join(self.model, "config.json")) as f:
                self.hf_config = json.load(f)
        self.hf_config = normalise_hf_config(self.hf_config)
        self.max_model_len = min(self.max_model_len, self.hf_config.max_position_embeddings)
        assert self.max_num_batched_tokens >= self.max_model_len
