#!/bin/bash
# r05, call B2: N > 1 rehearsals over gloo on one device, the two-process one-shot all-reduce, rocprofv3 kernel stats of the C3 and
# C2 decode loops.   bash tools/final_artifacts_b2.sh r05
set -o pipefail
RN=${1:-r05}
R=$PWD
OUT=$R/gpurun_out/$RN
mkdir -p $OUT
export TMPDIR=/tmp
GLM_BENCH_BACKEND=gloo GLM_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --layers 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_selfspawn_gpus2_gloo_rehearsal.json 2> $OUT/bench_selfspawn.err || { tail -5 $OUT/bench_selfspawn.err; exit 1; }
GLM_BENCH_BACKEND=gloo GLM_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --expert-parallel --batch 256 --layers 3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_selfspawn_gpus2_ep_gloo_rehearsal.json 2> $OUT/bench_selfspawn_ep.err || { tail -5 $OUT/bench_selfspawn_ep.err; exit 1; }
python - <<PY
import json
for f in ("bench_selfspawn_gpus2_gloo_rehearsal.json", "bench_selfspawn_gpus2_ep_gloo_rehearsal.json"):
    d = json.loads(open("$OUT/" + f).read().strip().splitlines()[-1])
    print(f, "collective:", json.dumps(d.get("collective")), "traffic attached:", "traffic" in (d.get("roofline") or {}), (d.get("roofline") or {}).get("traffic_note"))
PY
timeout -k 10 300 python tools/oneshot_allreduce_2proc.py > $OUT/oneshot_allreduce_2proc_graph_replay.log 2>&1 || { tail -5 $OUT/oneshot_allreduce_2proc_graph_replay.log; exit 1; }
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c2 --no-extras --no-graph --kernel-iters 5 > /tmp/prof_c3.log 2>&1 || { tail -5 /tmp/prof_c3.log; exit 1; }
cp "$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1)" $OUT/decode_c3_L46_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python3 $R/bench.py --batch 1 --ctx 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-c2 --no-extras --no-graph --kernel-iters 5 > /tmp/prof_c2.log 2>&1 || { tail -5 /tmp/prof_c2.log; exit 1; }
cp "$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1)" $OUT/decode_c2_L46_kernel_stats.csv
GLM4MOE_T1_ENGINE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2e -- python3 $R/bench.py --batch 1 --ctx 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-c2 --no-extras --no-graph --kernel-iters 5 > /tmp/prof_c2e.log 2>&1 || { tail -5 /tmp/prof_c2e.log; exit 1; }
cp "$(find /tmp/prof_c2e -name '*kernel_stats.csv' | head -1)" $OUT/decode_c2_L46_engine_kernel_stats.csv
echo "artefacts in $OUT:"; ls $OUT
