#!/bin/bash
# PMC passes over the prefill attention kernel (tools/prefill_attn_bench.py); run on the GPU box:
#   bash tools/pmc_prefill_attn.sh
# Each pass is its own rocprofv3 run with --kernel-trace only (no sys/runtime tracing).
set -e
R=$PWD
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_fa
cd /tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fa/$tag -- python3 $R/tools/prefill_attn_bench.py > $R/gpurun_out/pmc_fa/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$R/gpurun_out/pmc_fa/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "prefill_attn" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:32s} {tot[k]/n[k]:.4g}  (avg over {n[k]} launches)")
PY
