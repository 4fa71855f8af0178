#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c4; mkdir -p $O
timeout 600 python tests/ref_real_driver.py --mode train --iters 50 > $O/train.txt 2>&1; grep "train-growth" $O/train.txt | cut -c1-1500; tail -3 $O/train.txt | cut -c1-600
E=skyfall-gs_amd/sfgs/_exp
ROUNDS=2 tools/ab.sh skyfall-gs_amd/sfgs/libsfgs.so $E/lib_preocc6.so $E/lib_preocc8.so -- --steps 60 --warmup 10 > $O/ab_preocc.txt 2>&1; cat $O/ab_preocc.txt
