#!/bin/bash
# PMC passes over the prefill-shape grouped GEMMs (tools/prefill_gemm_only.py, T = 16384): HBM traffic, matrix-pipe
# busy cycles, wave wait buckets, LDS activity, clock under load.  Separate rocprofv3 --pmc passes with --kernel-trace
# only (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Run on the GPU box: bash tools/pmc_prefill_gemm.sh
# Writes gpurun_out/pmc_prefill_gemm${PG_MODE:+_$PG_MODE}.json (copy it to profiles/rNN/).  PG_MODE=a8: the W8A8 pair (r05).
set -e
R=$PWD
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_pg${PG_MODE:+_$PG_MODE}
rm -rf $OUT
mkdir -p $OUT
cd /tmp
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/prefill_gemm_only.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, json, statistics, collections
names = {"gemm_largem_kernel<1,": "grouped_gemm_gateup_silu@prefill16k", "gemm_largem_kernel<2,": "grouped_gemm_down@prefill16k"}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        for pat, key in names.items():
            if pat in r["Kernel_Name"]:
                vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "End_Timestamp" in r and "Start_Timestamp" in r:
                    dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
out = {}
for key, d in vals.items():
    m = {c: statistics.median(v) for c, v in d.items()}
    rec = {"counters_median": m, "launches": len(next(iter(d.values())))}
    if dur[key]:
        rec["duration_ns_median_profiled"] = statistics.median(dur[key])
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        rec["hbm_bytes_corrected"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
    if "GRBM_GUI_ACTIVE" in m and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0            # per-XCD active cycles (the counter sums the 8 XCDs)
        rec["mfma_pipe_busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 256 * 4)
        if "SQ_INSTS_MFMA" in m:
            rec["mfma_busy_cycles_per_mfma_inst"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / max(m["SQ_INSTS_MFMA"], 1.0)
        if dur[key]:
            rec["clock_ghz_under_load"] = cyc / rec["duration_ns_median_profiled"]
    out[key] = rec
import sys
sys.path.insert(0, "$R")
import bench
out["_meta"] = {"kernel_source_digest": bench._kernel_source_digest(), "git_head": bench._git("rev-parse", "--short", "HEAD"),
                "command": "tools/prefill_gemm_only.py under 4 rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ set | GRBM + instruction counts)",
                "notes": "FETCH_SIZE x2 (gfx950 counts 64 B per 128-B request), KB x 1024; pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)"}
out["_meta"]["mode"] = "${PG_MODE:-bf16}"
json.dump(out, open("$R/gpurun_out/pmc_prefill_gemm${PG_MODE:+_$PG_MODE}.json", "w"), indent=1)
for k, v in out.items():
    if k != "_meta":
        print(k, {x: v.get(x) for x in ("hbm_bytes_corrected", "mfma_pipe_busy_frac", "clock_ghz_under_load")})
PY
