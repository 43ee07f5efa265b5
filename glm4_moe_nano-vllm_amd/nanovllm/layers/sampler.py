"""Token sampler — reference: layers/sampler.py:5-20 (greedy when every temperature is 0,
otherwise temperature softmax + Gumbel-max).  Plain torch ops: not on the hot path."""
from __future__ import annotations

import torch
from torch import nn


class Sampler(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, logits: torch.Tensor, temperatures: torch.Tensor):
        if torch.all(temperatures == 0):
            return torch.argmax(logits, dim=-1)
        logits = logits.float().div_(temperatures.unsqueeze(dim=1))
        probs = torch.softmax(logits, dim=-1)
        noise = torch.empty_like(probs).exponential_(1).clamp_min_(1e-10)
        return probs.div_(noise).argmax(dim=-1)
