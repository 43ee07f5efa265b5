#!/bin/bash
# PMC passes over TWO prefill attention kernels at 4 x 4096 tokens — the default loop (k0) and an opt-in one (k1) selected by the
# environment variable named in PMC_FA_KNOB (default GLM_FA_MFMA32 = the 32x32x16 kernel; r04: GLM_FA_ROWS64 = 64 rows per wave):
# wave-cycle buckets, matrix-pipe busy, VALU / LDS instruction activity, LDS bank conflicts.  Each pass is its own rocprofv3
# run with --kernel-trace only.   bash tools/pmc_prefill_attn_ab.sh  ->  gpurun_out/pmc_fa_ab/summary.txt
set -e
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_fa_ab
mkdir -p $OUT
cd /tmp
KNOB=${PMC_FA_KNOB:-GLM_FA_MFMA32}
LABEL=${PMC_FA_LABEL:-32x32x16 (opt-in)}
# r05: PMC_FA_LIB0 / PMC_FA_LIB1 = two LIBRARIES instead of a knob (e.g. lib/libglm4moe_hip_r04.so built with -DGLM_FA_R05=0
# -DGLM_FA_STORE16=0 against the product library): k0 runs on LIB0, k1 on LIB1
for k in 0 1; do
  if [ -n "$PMC_FA_LIB1" ]; then
    if [ $k = 0 ]; then export GLM4MOE_HIP_LIB=$PMC_FA_LIB0; else export GLM4MOE_HIP_LIB=$PMC_FA_LIB1; fi
  else
    export $KNOB=$k
  fi
  for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
    tag=k${k}_$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/tools/prefill_attn_bench.py > $OUT/$tag.log 2>&1 || { tail -3 $OUT/$tag.log; }
  done
done
python3 - <<PY | tee $OUT/summary.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob("$OUT/k*/*/*counter_collection.csv"):
    k = f.split("/pmc_fa_ab/")[1][:2]
    for r in csv.DictReader(open(f)):
        if "prefill_attn" in r["Kernel_Name"]:
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
names = sorted(set(tot["k0"]) | set(tot["k1"]))
print(f"{'counter (average per launch)':34s} {'default kernel':>20s} {'$LABEL':>20s}")
for c in names:
    a = tot["k0"][c] / max(1, n["k0"][c]); b = tot["k1"][c] / max(1, n["k1"][c])
    print(f"{c:34s} {a:20.4g} {b:20.4g}")
PY
rm -rf $OUT/k*
