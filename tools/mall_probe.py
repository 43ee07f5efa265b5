"""Infinity-Cache (256 MiB) read bandwidth against HBM: a buffer of S MB is read repeatedly (resident after the first
pass when S fits) vs a 2 GiB buffer streamed once per pass.  Reader: torch's elementwise copy into a small ring is not a
pure read, so the probe uses bf16 .sum() (a read-only reduction) and reports GB/s of bytes read."""
import torch, sys
def bw(nbytes, passes=20):
    x = torch.empty(nbytes // 2, dtype=torch.bfloat16, device="cuda").normal_()
    torch.cuda.synchronize()
    x.sum(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(passes): x.sum()
    b.record(); torch.cuda.synchronize()
    return nbytes * passes / (a.elapsed_time(b) * 1e-3) / 1e9
for mb in (16, 64, 128, 192, 512, 2048):
    print(f"{mb:5d} MB re-read: {bw(mb << 20):8.0f} GB/s", flush=True)
