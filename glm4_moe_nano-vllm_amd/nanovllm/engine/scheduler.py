"""Reference module path `nanovllm.engine.scheduler` (implementation: engine/core.py)."""
from nanovllm.engine.core import Scheduler  # noqa: F401
