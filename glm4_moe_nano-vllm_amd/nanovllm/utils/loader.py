"""safetensors helpers shared by the per-module load_weights methods (reference: each module
rescans the shards itself, e.g. models/glm4_moe/mlp.py:83-131; utils/loader.py:12-28)."""
from __future__ import annotations

import os
from glob import glob

from safetensors import safe_open


def shard_files(path: str) -> list[str]:
    if os.path.isdir(path):
        return sorted(glob(os.path.join(path, "*.safetensors")))
    return [path]


def load_prefixed(path: str, prefix: str) -> dict:
    """All tensors whose key starts with `prefix`, from a directory of shards or one file."""
    out = {}
    for f in shard_files(path):
        with safe_open(f, framework="pt", device="cpu") as sf:
            for key in sf.keys():
                if key.startswith(prefix):
                    out[key] = sf.get_tensor(key)
    return out


def load_one(path: str, key: str):
    for f in shard_files(path):
        with safe_open(f, framework="pt", device="cpu") as sf:
            if key in sf.keys():
                return sf.get_tensor(key)
    return None
