"""Checkpoint lookup helpers (nanovllm/utils/loader.py): whole-component prefixes and the key -> shard index."""
import json
import os
import sys

import torch
from safetensors.torch import save_file

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))


def test_prefix_matches_whole_components_and_uses_the_shard_index(tmp_path):
    from nanovllm.utils import loader
    p = "model.layers.1.mlp.experts"
    # shard B sorts first and holds experts.10, shard A holds experts.1: a startswith() scan would pick the wrong expert
    save_file({f"{p}.10.gate_proj.weight": torch.full((2, 2), 10.0), f"{p}.11.gate_proj.weight": torch.full((2, 2), 11.0)},
              str(tmp_path / "a-00001.safetensors"))
    save_file({f"{p}.1.gate_proj.weight": torch.full((2, 2), 1.0), "model.norm.weight": torch.ones(2)},
              str(tmp_path / "b-00002.safetensors"))
    d = str(tmp_path)
    loader._key_index.cache_clear()
    sd = loader.load_prefixed(d, f"{p}.1")
    assert list(sd) == [f"{p}.1.gate_proj.weight"] and float(sd[f"{p}.1.gate_proj.weight"][0, 0]) == 1.0
    assert float(loader.load_one(d, f"{p}.10.gate_proj.weight")[0, 0]) == 10.0
    assert loader.load_one(d, f"{p}.2.gate_proj.weight") is None
    assert set(loader.load_prefixed(d, p)) == {f"{p}.1.gate_proj.weight", f"{p}.10.gate_proj.weight", f"{p}.11.gate_proj.weight"}
    # with an index file the weight map decides which shard is opened
    with open(tmp_path / "model.safetensors.index.json", "w") as f:
        json.dump({"weight_map": {"model.norm.weight": "b-00002.safetensors"}}, f)
    loader._key_index.cache_clear()
    assert loader.load_one(d, "model.norm.weight") is not None
    assert loader.load_one(d, f"{p}.1.gate_proj.weight") is None  # not in the map: the map is authoritative
    loader._key_index.cache_clear()
