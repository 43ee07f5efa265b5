"""r05: where the persistent prefill attention (option fa_persist) spends a slot's time, and how evenly the slots finish.
Needs the stamp variant: python glm4_moe_nano-vllm_amd/build.py --variant _fapst GLM_STUDY_KERNELS=1 GLM_FAP_STAMP=1 ;
GLM4MOE_HIP_LIB=.../libglm4moe_hip_fapst.so GLM_FA_PERSIST=1 python tools/fap_stamps.py [B S]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "glm4_moe_nano-vllm_amd"))
from nanovllm import hip_ops as ops
B, S = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4, 4096)
nh, nkv, hd = 96, 8, 128
q = torch.randn(B * S, nh, hd, device="cuda").to(torch.bfloat16)
k = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
v = torch.randn(B * S, nkv, hd, device="cuda").to(torch.bfloat16)
cu = torch.arange(0, (B + 1) * S, S, dtype=torch.int32, device="cuda")
for _ in range(3): ops.varlen_attn_prefill(q, k, v, cu, cu, S, hd ** -0.5)
torch.cuda.synchronize()
lib = ops.load_library()
f = lib.glm_debug_fap_stamps; f.restype = ctypes.c_int; f.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
buf = (ctypes.c_ulonglong * (1024 * 8))(); assert f(buf) == 0
t = torch.tensor(list(buf), dtype=torch.float64).view(1024, 8)[:512]
t0 = t[:, 0].min()
start, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0   # us (100 MHz)
print(f"{B}x{S}: slots start {start.min():.1f}..{start.max():.1f} us, finish min {end.min():.1f} median {end.median():.1f} max {end.max():.1f} us")
order = end.argsort()
print("  earliest slots:", [(int(i), round(float(end[i]), 1)) for i in order[:6]])
print("  latest slots:  ", [(int(i), round(float(end[i]), 1)) for i in order[-6:]])
for x in range(8):
    e = end[x::8]
    print(f"  XCD {x}: finish min {e.min():.1f} median {e.median():.1f} max {e.max():.1f}")
hw = [int(x) for x in torch.tensor(list(buf), dtype=torch.float64).view(1024, 8)[:512, 7].tolist()]
raw = list(buf)
place = []
for w in range(512):
    word = raw[w * 8 + 7]; hwid, xcc = word & 0xffffffff, (word >> 32) & 0xf
    place.append((xcc, (hwid >> 13) & 7, (hwid >> 12) & 1, (hwid >> 8) & 15))  # xcc, se, sh, cu
from collections import defaultdict
bycu = defaultdict(list)
for w in range(512): bycu[place[w]].append(w)
print(f"  placement: {len(bycu)} distinct (xcc, se, sh, cu); workgroups per CU: {sorted(set(len(v) for v in bycu.values()))}")
print("  slots sharing a CU finish (us):", [(v, [round(float(end[i])) for i in v]) for v in list(bycu.values())[:6]])
xccs = defaultdict(list)
for w in range(512): xccs[(w % 8)].append(place[w][0])
print("  blockIdx % 8 -> XCC ids seen:", {k: sorted(set(v)) for k, v in xccs.items()})
h = torch.histc(end, bins=12, min=float(end.min()), max=float(end.max()))
print("  finish histogram:", [int(x) for x in h.tolist()], f"bins of {(float(end.max()) - float(end.min())) / 12:.0f} us from {float(end.min()):.0f}")
slow = [int(i) for i in order[-40:]]
print("  40 latest slots: (slot, xcc, se, sh, cu):", [(i,) + place[i] for i in slow])
print("  interior-tile cycles: fastest slot", float(t[order[0], 5]), "slowest", float(t[order[-1], 5]), "; general:", float(t[order[0], 4]), float(t[order[-1], 4]))
names = {2: "blocks", 3: "query-row wait", 4: "general tiles", 5: "interior tiles", 6: "output rows + switch"}
tot = t[:, 3:7].sum(1)
for i, n in names.items():
    c = t[:, i]
    if i == 2: print(f"  {n}: {c.min():.0f}..{c.max():.0f}")
    else: print(f"  {n:22s} cycles/slot median {c.median():12.0f}  ({100 * c.median() / tot.median():.1f} %)   per block {c.median() / t[:, 2].median():9.0f}")
