#!/bin/bash
# round 5 (gpurun): composite_bwd in its own translation unit under the max-ILP scheduling strategy -- raster parity tests,
# bit-for-bit comparison with the previous build (tools/bitcompare.py), then an alternating A/B of bench.py
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests -q -m gpu -x -k "raster or robust or parity or reference_real or training_loop or joint" 2>&1 | tail -3 | tee gpurun_out/sched_tests.txt
timeout 300 python tools/bitcompare.py > gpurun_out/bits_main.json 2>/dev/null; SFGS_LIB=$PWD/$1 timeout 300 python tools/bitcompare.py > gpurun_out/bits_prev.json 2>/dev/null
python tools/bitcompare.py --diff gpurun_out/bits_main.json gpurun_out/bits_prev.json 2>&1 | tail -6 | tee gpurun_out/sched_bitcompare.txt
ROUNDS=3 bash tools/ab.sh skyfall-gs_amd/sfgs/libsfgs.so "$1" | tee gpurun_out/sched_ab.txt
