import os, sys, time, torch
import torch.nn.functional as F
sys.path.insert(0, '.')
from oracle import glm4_moe_oracle as O
print("cpus", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
def t(fn, n=2):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    return (time.perf_counter() - t0) / n
bf = torch.bfloat16
x32 = torch.randn(32, 4096).to(bf); x4 = torch.randn(4, 4096).to(bf)
wq = torch.randn(14336, 4096).to(bf); we = torch.randn(2816, 4096).to(bf); wd = torch.randn(4096, 1408).to(bf)
a4 = torch.randn(4, 1408).to(bf)
for nt in (16, 64, 256):
    torch.set_num_threads(nt)
    print(nt, "qkv 32x14336x4096: %.4f s" % t(lambda: F.linear(x32, wq)),
          "expert gu 4x2816x4096: %.4f s" % t(lambda: F.linear(x4, we)),
          "expert down 4x4096x1408: %.4f s" % t(lambda: F.linear(a4, wd)), flush=True)
q = torch.randn(1, 96, 128).to(bf); k = torch.randn(4096, 8, 128).to(bf); v = torch.randn(4096, 8, 128).to(bf)
for nt in (16, 64, 256):
    torch.set_num_threads(nt)
    print(nt, "attn 1 seq: %.4f s" % t(lambda: O._attend(q, k, v, 0.088, None)), flush=True)
xf = x32.float(); wqf = wq.float()
torch.set_num_threads(64)
print("fp32 qkv: %.4f s" % t(lambda: F.linear(xf, wqf)))
