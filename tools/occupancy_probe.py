"""Ask the HIP runtime how many prefill-attention workgroups fit on a CU for a few LDS sizes."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
torch.zeros(1, device="cuda")
lib = ops.load_library()
f = lib.glm_debug_prefill_attn_occupancy
f.restype = ctypes.c_int; f.argtypes = [ctypes.c_int]
for kb in (16, 32, 48, 64, 72, 80):
    print(f"dynamic LDS {kb} KiB -> {f(kb * 1024)} workgroups/CU")
p = torch.cuda.get_device_properties(0)
print("shared_memory_per_block", p.shared_memory_per_block, "per_multiprocessor", getattr(p, "shared_memory_per_multiprocessor", None), "regs_per_multiprocessor", getattr(p, "regs_per_multiprocessor", None))
