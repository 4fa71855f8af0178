#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c7; mkdir -p $O
E=skyfall-gs_amd/sfgs/_exp
ROUNDS=2 tools/ab.sh $E/lib_base.so $E/lib_mfma1.so $E/lib_mfma2.so -- --steps 60 --warmup 10 > $O/ab_mfma.txt 2>&1; cat $O/ab_mfma.txt
