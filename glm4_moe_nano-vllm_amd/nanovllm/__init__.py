"""nanovllm — MI355X-native drop-in for the GLM-4-MoE per-layer forward path of nano-vllm."""
