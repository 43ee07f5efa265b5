"""`nanovllm.LLM` — reference: llm.py:4-5."""
from nanovllm.engine.llm_engine import LLMEngine


class LLM(LLMEngine):
    pass
