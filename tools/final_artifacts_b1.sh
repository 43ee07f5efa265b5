#!/bin/bash
# r05, call B1 of the round artefacts (the suite is call A, the rocprof kernel stats and rehearsals call B2):
# PMC traffic of the decode hot kernels, the bench line, PMC of the W8A8 GEMM pair.   bash tools/final_artifacts_b1.sh r05
set -o pipefail
RN=${1:-r05}
R=$PWD
OUT=$R/gpurun_out/$RN
mkdir -p $OUT $R/profiles/$RN
export TMPDIR=/tmp
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1 || { tail -20 $OUT/pmc_traffic.txt; exit 1; }
cp gpurun_out/pmc_traffic.json $OUT/pmc_traffic.json
cp gpurun_out/pmc_traffic.json $R/profiles/$RN/pmc_traffic.json
rm -rf gpurun_out/pmc_tr
cd $R
python bench.py > $OUT/bench_L46.json 2> $OUT/bench_L46.err || { tail -5 $OUT/bench_L46.err; exit 1; }
python tools/show_bench.py $OUT/bench_L46.json
PG_MODE=a8 timeout -k 10 280 bash tools/pmc_prefill_gemm.sh > $OUT/pmc_a8.log 2>&1; tail -3 $OUT/pmc_a8.log
cp gpurun_out/pmc_prefill_gemm_a8.json $OUT/ 2>/dev/null
echo "artefacts in $OUT:"; ls $OUT
