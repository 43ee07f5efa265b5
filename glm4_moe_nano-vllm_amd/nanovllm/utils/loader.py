"""safetensors helpers shared by the per-module load_weights methods (reference: each module
rescans the shards itself, e.g. models/glm4_moe/mlp.py:83-131; utils/loader.py:12-28).

A checkpoint directory is scanned ONCE: the key -> shard index comes from model.safetensors.index.json when present,
otherwise from one pass over the shard headers; lookups then open only the shard that holds the tensor.  A prefix
matches whole dotted components ("...experts.1" does not match "...experts.10"; the reference's startswith() does,
with last-match-wins, mlp.py:86-111)."""
from __future__ import annotations

import json
import os
from functools import lru_cache
from glob import glob

from safetensors import safe_open


def shard_files(path: str) -> list[str]:
    if os.path.isdir(path):
        return sorted(glob(os.path.join(path, "*.safetensors")))
    return [path]


@lru_cache(maxsize=4)
def _key_index(path: str) -> dict:
    """key -> shard file."""
    idx_file = os.path.join(path, "model.safetensors.index.json") if os.path.isdir(path) else None
    if idx_file and os.path.exists(idx_file):
        with open(idx_file) as f:
            wm = json.load(f).get("weight_map", {})
        if wm:
            return {k: os.path.join(path, v) for k, v in wm.items()}
    out = {}
    for f in shard_files(path):
        with safe_open(f, framework="pt", device="cpu") as sf:
            for key in sf.keys():
                out[key] = f
    return out


def _fetch(files_to_keys: dict) -> dict:
    out = {}
    for f, keys in files_to_keys.items():
        with safe_open(f, framework="pt", device="cpu") as sf:
            for k in keys:
                out[k] = sf.get_tensor(k)
    return out


def load_prefixed(path: str, prefix: str) -> dict:
    """All tensors under `prefix` (the key itself or `prefix.`-children), from a directory of shards or one file."""
    by_file: dict = {}
    dotted = prefix + "."
    for key, f in _key_index(path).items():
        if key == prefix or key.startswith(dotted):
            by_file.setdefault(f, []).append(key)
    return _fetch(by_file)


def load_one(path: str, key: str):
    f = _key_index(path).get(key)
    if f is None:
        return None
    return _fetch({f: [key]})[key]
