#!/bin/bash
# SQ-counter calibration, runs on the GPU box: tools/microbench/issue_bench -- one instruction form
# per kernel at 4 waves per SIMD -- under the same PMC passes as tools/collect_profiles.sh. The pure v_fma_f32 loop
# saturates the VALU pipe and the ds_read_b64 loop the CU's LDS pipe: the factors that make their derived ratios read
# 1.00 go to gpurun_out/sq_cal/calibration.json (copied to profiles/r3_sq_calibration.json by the author).
set -u
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD/gpurun_out/sq_cal; mkdir -p "$R"
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
BIN=tools/microbench/issue_bench
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/issue_bench.hip -o $BIN
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 -d "$R/sq1" -o sq1 -- $BIN > "$R/sq1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 -d "$R/sq2" -o sq2 -- $BIN > "$R/sq2.log" 2>&1
python tools/prof_summary.py --all-kernels --pmc "$R/sq1/sq1_results.db" "$R/sq2/sq2_results.db" --sq-json "$R/micro_sq.json" > "$R/micro_summary.txt" 2>&1
python - <<PY
import json
d = json.load(open("$R/micro_sq.json"))
fma, lds = d["k_fma"], d["k_ds_read_b64_row"]
cal = {"factors": {"valu_util": 1.0 / fma["valu_util_raw"], "lds_busy": 1.0 / lds["lds_busy_raw"]},
       "evidence": {"k_fma": {k: fma[k] for k in ("valu_util_raw", "valu_of_wave", "occupancy_waves_per_simd", "avg_us")},
                    "k_ds_read_b64_row": {k: lds[k] for k in ("lds_busy_raw", "valu_util_raw", "occupancy_waves_per_simd", "avg_us")}},
       "all": {k: {m: v[m] for m in ("valu_util_raw", "lds_busy_raw", "occupancy_waves_per_simd")} for k, v in d.items()}}
json.dump(cal, open("$R/calibration.json", "w"), indent=1)
print(json.dumps(cal["factors"]), json.dumps(cal["evidence"]))
PY
rm -rf "$R/sq1" "$R/sq2"
