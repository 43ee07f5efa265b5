#!/bin/bash
# Round artefacts on the GPU box, in one call:  bash tools/final_artifacts.sh r02
#   1. the whole GPU test suite                      -> gpurun_out/<r>/gputests.log
#   2. PMC traffic of the hot kernels (two --pmc passes) -> gpurun_out/<r>/pmc_traffic.json (also placed under profiles/<r>/ of
#      the box's copy, so that the bench lines below attach it: same kernel sources, digest checked by bench.py)
#   3. bench.py (C3 headline; C2, fp8 experts and the prefill-side figures under `extra`)
#   4. rocprofv3 kernel-trace stats of the C3 and the C2 decode loops (csv summaries only; traces stay in /tmp)
# Copy gpurun_out/<r>/* into profiles/<r>/ afterwards.
set -o pipefail
RN=${1:-r04}
R=$PWD
OUT=$R/gpurun_out/$RN
mkdir -p $OUT $R/profiles/$RN
export TMPDIR=/tmp
if [ -z "$SKIP_SUITE" ]; then   # (r05: the suite and the rest are two calls — together they pass the 20-minute limit of a call)
python -m pytest tests -q -m gpu > $OUT/gputests.log 2>&1 || { tail -20 $OUT/gputests.log; exit 1; }
tail -2 $OUT/gputests.log
cp gpurun_out/parity_fullsize.jsonl $OUT/ 2>/dev/null
fi
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1 || { tail -20 $OUT/pmc_traffic.txt; exit 1; }
cp gpurun_out/pmc_traffic.json $OUT/pmc_traffic.json
cp gpurun_out/pmc_traffic.json $R/profiles/$RN/pmc_traffic.json
rm -rf gpurun_out/pmc_tr
python bench.py > $OUT/bench_L46.json 2> $OUT/bench_L46.err || { tail -5 $OUT/bench_L46.err; exit 1; }
python tools/show_bench.py $OUT/bench_L46.json
GLM_BENCH_BACKEND=gloo GLM_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --layers 4 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_selfspawn_gpus2_gloo_rehearsal.json 2> $OUT/bench_selfspawn.err || { tail -5 $OUT/bench_selfspawn.err; exit 1; }
GLM_BENCH_BACKEND=gloo GLM_BENCH_SINGLE_DEVICE=1 python bench.py --gpus 2 --expert-parallel --batch 256 --layers 3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_selfspawn_gpus2_ep_gloo_rehearsal.json 2> $OUT/bench_selfspawn_ep.err || { tail -5 $OUT/bench_selfspawn_ep.err; exit 1; }
timeout -k 10 300 python tools/oneshot_allreduce_2proc.py > $OUT/oneshot_allreduce_2proc_graph_replay.log 2>&1 || { tail -5 $OUT/oneshot_allreduce_2proc_graph_replay.log; exit 1; }
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c2 --no-extras --no-graph --kernel-iters 5 > /tmp/prof_c3.log 2>&1 || { tail -5 /tmp/prof_c3.log; exit 1; }
cp "$(find /tmp/prof_c3 -name '*kernel_stats.csv' | head -1)" $OUT/decode_c3_L46_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -- python3 $R/bench.py --batch 1 --ctx 1024 --steps 20 --warmup 3 --no-cpu-baseline --no-c2 --no-extras --no-graph --kernel-iters 5 > /tmp/prof_c2.log 2>&1 || { tail -5 /tmp/prof_c2.log; exit 1; }
cp "$(find /tmp/prof_c2 -name '*kernel_stats.csv' | head -1)" $OUT/decode_c2_L46_kernel_stats.csv
echo "artefacts in $OUT:"; ls -la $OUT
