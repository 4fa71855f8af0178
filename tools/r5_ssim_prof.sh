#!/bin/bash
# round 5 (gpurun): where does fused_ssim's time go? kernel trace + two SQ passes of tools/bench_aux.py
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
R=$PWD/gpurun_out/prof_r5_ssim; mkdir -p $R
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
timeout 300 python tools/bench_aux.py > $R/bench_aux.json 2>/dev/null; cat $R/bench_aux.json
timeout 300 rocprofv3 --kernel-trace --stats -d $R/kt -o kt -- python tools/bench_aux.py > $R/kt.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 -d $R/sq1 -o sq1 -- python tools/bench_aux.py > $R/sq1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 -d $R/sq2 -o sq2 -- python tools/bench_aux.py > $R/sq2.log 2>&1
python tools/prof_summary.py --by-grid ssim --cal profiles/r3_sq_calibration.json $R/kt/kt_results.db --pmc $R/sq1/sq1_results.db $R/sq2/sq2_results.db --sq-json $R/sq.json > $R/summary.txt 2>&1
rm -rf $R/kt $R/sq1 $R/sq2 $R/*.log
grep -i "ssim" $R/summary.txt | cut -c1-220 | head -24
