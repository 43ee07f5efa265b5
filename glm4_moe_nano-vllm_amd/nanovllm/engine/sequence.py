"""Reference module path `nanovllm.engine.sequence` (implementation: engine/core.py)."""
from nanovllm.engine.core import Sequence, SequenceStatus  # noqa: F401
