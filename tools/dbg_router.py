import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'glm4_moe_nano-vllm_amd')
from nanovllm import hip_ops as ops
from oracle import glm4_moe_oracle as O
from util import load_golden
t, meta = load_golden("router")
for tag, m in meta.items():
    x, gw, bias = t[f"{tag}.x"], t[f"{tag}.gate_w"], t[f"{tag}.bias"]
    idx, w, logits = ops.router_topk(x.cuda(), gw.cuda(), bias.cuda(), m["k"], m["rsf"], return_logits=True)
    rl, rs, rc = O.router_scores(x, gw, bias)
    exact = x.double() @ gw.double().T
    d = (logits.cpu() != rl)
    print(tag, "logit mismatches:", int(d.sum()), "of", d.numel())
    for (r, c) in d.nonzero().tolist():
        print("   row", r, "col", c, "gpu", float(logits[r, c]), "cpu", float(rl[r, c]), "exact", float(exact[r, c]))
    ridx = t[f"{tag}.idx"]
    for r in range(x.shape[0]):
        if set(idx[r].tolist()) != set(ridx[r].tolist()):
            print("   set mismatch row", r, sorted(idx[r].tolist()), sorted(ridx[r].tolist()))
