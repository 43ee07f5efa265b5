"""Reference module path `nanovllm.engine.block_manager` (implementation: engine/core.py)."""
from nanovllm.engine.core import BlockManager  # noqa: F401
