#!/bin/bash
# Runs on the GPU box (via gpurun): checks an experiment against a previous build on ONE box.
#   tools/ab_check.sh <out-tag> <prev.so> [more variant .so ...]
# 1. raster parity tests on the shipped library; 2. bit-for-bit comparison of every output of shipped vs prev
# (tools/bitcompare.py); 3. alternating bench.py A/B of shipped, prev and the extra variants (tools/ab.sh);
# 4. wave timeline if skyfall-gs_amd/sfgs/_exp/lib_timeline.so exists.  SKIP_TESTS=1 skips step 1.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
TAG=$1; shift; PREV=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 1500 python -m pytest tests -q -m gpu -x -k "${TESTS_K:-raster or robust or parity}" 2>&1 | tail -3 | tee $O/tests.txt
fi
timeout 300 python tools/bitcompare.py > $O/bits_main.json 2>/dev/null
SFGS_LIB=$PWD/$PREV timeout 300 python tools/bitcompare.py > $O/bits_prev.json 2>/dev/null
python tools/bitcompare.py --diff $O/bits_main.json $O/bits_prev.json 2>&1 | tail -3 | tee $O/bitcompare.txt
ROUNDS=${ROUNDS:-3} bash tools/ab.sh skyfall-gs_amd/sfgs/libsfgs.so $PREV "$@" -- ${BENCH_ARGS:-} | tee $O/ab.txt
if [ -f skyfall-gs_amd/sfgs/_exp/lib_timeline.so ]; then
  SFGS_LIB=$PWD/skyfall-gs_amd/sfgs/_exp/lib_timeline.so timeout 300 python tools/timeline.py 2>/dev/null | tee $O/timeline.json | cut -c1-700
fi
