#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace, the two HBM PMC passes (separate runs, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass) and three SQ PMC passes (8 SQ
# slots per pass) of bench.py, and summarises them (tools/prof_summary.py: per-kernel duration, HBM traffic, VALU
# utilisation, LDS stall share, achieved occupancy).
# usage: tools/collect_profiles.sh <tag> [passes]   -> gpurun_out/prof_<tag>/{summary.txt,traffic.json,sq.json}
#   passes: any of "kt fetch write sq" (default: all)
set -u
TAG=${1:-run}
PASSES=${2:-"kt fetch write sq"}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_$TAG
mkdir -p "$R"
ARGS="--steps 5 --warmup 2 --cpu-sample 0 ${BENCH_EXTRA:-}"
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
SQ3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LEVEL_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL"
DBS=""
for p in $PASSES; do
  case $p in
    kt) timeout 300 rocprofv3 --kernel-trace --stats -d "$R/kt" -o kt -- python bench.py $ARGS > "$R/kt.log" 2>&1 ;;
    fetch) timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$R/fetch" -o fetch -- python bench.py $ARGS > "$R/fetch.log" 2>&1; DBS="$DBS $R/fetch/fetch_results.db" ;;
    write) timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$R/write" -o write -- python bench.py $ARGS > "$R/write.log" 2>&1; DBS="$DBS $R/write/write_results.db" ;;
    sq2) timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 -d "$R/sq2" -o sq2 -- python bench.py $ARGS > "$R/sq2.log" 2>&1; DBS="$DBS $R/sq2/sq2_results.db" ;;
    sq)
      timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 -d "$R/sq1" -o sq1 -- python bench.py $ARGS > "$R/sq1.log" 2>&1; DBS="$DBS $R/sq1/sq1_results.db"
      timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 -d "$R/sq2" -o sq2 -- python bench.py $ARGS > "$R/sq2.log" 2>&1; DBS="$DBS $R/sq2/sq2_results.db"
      timeout 300 rocprofv3 --kernel-trace --pmc $SQ3 -d "$R/sq3" -o sq3 -- python bench.py $ARGS > "$R/sq3.log" 2>&1; DBS="$DBS $R/sq3/sq3_results.db" ;;
  esac
done
EXIST=""
for d in $DBS; do [ -f "$d" ] && EXIST="$EXIST $d"; done
KT=""; [ -f "$R/kt/kt_results.db" ] && KT="$R/kt/kt_results.db"
CAL=""; [ -f profiles/r3_sq_calibration.json ] && CAL="--cal profiles/r3_sq_calibration.json"   # tools/calibrate_sq.sh
python tools/prof_summary.py $CAL $KT --pmc $EXIST --json "$R/traffic.json" --sq-json "$R/sq.json" > "$R/summary.txt" 2>&1
grep '^{' "$R/kt.log" 2>/dev/null | tail -1 > "$R/bench_under_rocprof.json"
for f in "$R"/*.log; do tail -3 "$f" > "$f.tail"; rm -f "$f"; done
rm -rf "$R/kt" "$R/fetch" "$R/write" "$R/sq1" "$R/sq2" "$R/sq3"
tail -60 "$R/summary.txt"
