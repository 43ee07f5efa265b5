import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "tok/s", d["value"], "ms/step", d["ms_per_step"], "ms/layer", d["config"]["ms_per_layer"])
for k, v in d["kernels"].items():
    print(f"   {k:40s} {v['achieved']:8.1f} {v['unit']:8s} frac {v['frac']:.3f}  {v['kernel_us_mean']:9.1f} us (best {v.get('kernel_us_best', 0)})")
