"""SiLU-gated MLP used for the dense layer-0 MLP, every routed expert and the shared expert —
reference: models/glm4_moe/mlp.py:12-131.

forward = two HIP launches: gate|up GEMM with the SiLU·mul epilogue fused
(glm_grouped_gemm_gateup_silu, one group) and the down GEMM (glm_grouped_gemm_down); the
reference runs F.linear → chunk/silu/mul → F.linear."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from nanovllm import hip_ops
from nanovllm.distributed.parallel_state import get_tp_group
from nanovllm.layers.activation import SiluAndMul
from nanovllm.layers.linear import MergedColumnParallelLinear, RowParallelLinear
from nanovllm.utils.loader import load_one


class Glm4MoeMLP(nn.Module):

    def __init__(self, hidden_size: int, intermediate_size: int, hidden_act: str,
                 quant_config: Optional[object] = None, reduce_results: bool = True,
                 prefix: str = "") -> None:
        super().__init__()
        if quant_config is not None:
            raise NotImplementedError("quantised MLP is outside the MI355X hot-path scope")
        self.gate_up_proj = MergedColumnParallelLinear(hidden_size, [intermediate_size] * 2, bias=False)
        self.down_proj = RowParallelLinear(intermediate_size, hidden_size, bias=False)
        if hidden_act.lower() != "silu":
            raise ValueError(f"Unsupported activation: {hidden_act}. Only silu is supported for now.")
        self.act_fn = SiluAndMul()
        self.prefix = prefix
        self.reduce_results = reduce_results

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        T = x2.shape[0]
        act = hip_ops.grouped_gemm_gateup_silu(x2, self.gate_up_proj.weight.data.unsqueeze(0), None, None, T)
        y = hip_ops.grouped_gemm_down(act, self.down_proj.weight.data.unsqueeze(0), None, None, 0)
        if self.reduce_results and self.down_proj.tp_size > 1:
            get_tp_group().all_reduce(y)
        return y.view(*shape[:-1], y.shape[-1])

    @torch.no_grad()
    def forward_debug(self, x: torch.Tensor) -> dict:
        """Unfused evaluation returning the intermediates the reference names (mlp.py:56-80)."""
        out = {"x_in": x}
        gate_up = self.gate_up_proj(x)
        gate, up = gate_up.chunk(2, dim=-1)
        out.update(gate_up=gate_up, gate=gate, up=up)
        out["act_x"] = self.act_fn(gate_up)
        out["down"] = out["ret_x"] = self.down_proj(out["act_x"])
        return out

    def load_weights(self, model_dir: str, prefix: str):
        gate_w, up_w, down_w = (load_one(model_dir, f"{prefix}.{n}.weight") for n in ("gate_proj", "up_proj", "down_proj"))
        assert gate_w is not None and up_w is not None and down_w is not None, f"MLP weights missing under {prefix}"
        self.gate_up_proj.weight_loader(self.gate_up_proj.weight, gate_w, 0)
        self.gate_up_proj.weight_loader(self.gate_up_proj.weight, up_w, 1)
        self.down_proj.weight_loader(self.down_proj.weight, down_w)
