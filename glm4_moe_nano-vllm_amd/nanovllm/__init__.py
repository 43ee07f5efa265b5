"""nanovllm — MI355X-native drop-in for the GLM-4-MoE per-layer forward path of nano-vllm.

`LLM` / `SamplingParams` are imported lazily so that `import nanovllm.layers...` does not pull
the engine (and torch.multiprocessing) into kernel-only users."""


def __getattr__(name):
    if name == "LLM":
        from nanovllm.llm import LLM
        return LLM
    if name == "SamplingParams":
        from nanovllm.sampling_params import SamplingParams
        return SamplingParams
    raise AttributeError(name)
