"""Thin model wrapper around the hot path — reference: models/glm4_moe/model.py:14-145.

Differences that are configuration, not behaviour: the number of decoder layers is
`config.num_layers_to_load` when present, else `config.num_hidden_layers` (the reference
hard-codes 2 for debugging, model.py:21-22); `config.final_norm_without_residual` (default True)
keeps the reference's final `norm(hidden_states)` that ignores the residual (model.py:75-76)."""
from __future__ import annotations

import torch
import torch.nn as nn

from nanovllm.layers.embed_head import ParallelLMHead, VocabParallelEmbedding
from nanovllm.layers.layernorm import RMSNorm
from nanovllm.utils.loader import load_one
from .decode_layer import Glm4MoeDecoderLayer


class Glm4MoeModel(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_tokens = VocabParallelEmbedding(config.vocab_size, config.hidden_size)
        n_layers = getattr(config, "num_layers_to_load", None) or config.num_hidden_layers
        self.layers = nn.ModuleList(
            [Glm4MoeDecoderLayer(config, prefix=f"model.layers.{i}") for i in range(n_layers)])
        self.norm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.final_norm_without_residual = getattr(config, "final_norm_without_residual", True)

    def load_weights(self, model_path: str):
        w = load_one(model_path, "model.embed_tokens.weight")
        if w is None:
            raise ValueError(f"'model.embed_tokens.weight' not found in '{model_path}'")
        self.embed_tokens.weight_loader(self.embed_tokens.weight, w)
        for i, layer in enumerate(self.layers):
            layer.load_weights(model_path, prefix=f"model.layers.{i}")
        w = load_one(model_path, "model.norm.weight")
        if w is None:
            raise ValueError(f"'model.norm.weight' not found in '{model_path}'")
        self.norm.weight.data.copy_(w)

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        # a layer's MoE combine is folded into the add-RMSNorm that consumes it (next layer's input norm or the
        # final norm); RMSNorm materialises it where the reference's residual-free final norm is kept
        for layer in self.layers:
            hidden_states, residual = layer(positions, hidden_states, residual, defer_combine=True)
        if self.final_norm_without_residual or residual is None:
            return self.norm(hidden_states)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states


class Glm4MoeForCausalLM(nn.Module):
    packed_modules_mapping = {
        "qkv_proj": ("qkv_proj", "qkv"),
        "gate_proj": ("gate_up_proj", 0),
        "up_proj": ("gate_up_proj", 1),
    }

    def __init__(self, config):
        super().__init__()
        self.model = Glm4MoeModel(config)
        self.config = config
        self.quant_config = None
        self.lm_head = ParallelLMHead(config.vocab_size, config.hidden_size, quant_config=None)
        if getattr(config, "tie_word_embeddings", False):
            self.lm_head.weight.data = self.model.embed_tokens.weight.data

    def load_weights(self, model_path: str):
        self.model.load_weights(model_path)
        if not getattr(self.config, "tie_word_embeddings", False):
            w = load_one(model_path, "lm_head.weight")
            if w is not None:
                self.lm_head.weight_loader(self.lm_head.weight, w)
            else:
                print(f"warning: 'lm_head.weight' not found in '{model_path}'")

    def forward(self, input_ids: torch.Tensor, positions: torch.Tensor) -> torch.Tensor:
        return self.model(input_ids, positions)

    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        return self.lm_head(hidden_states)
