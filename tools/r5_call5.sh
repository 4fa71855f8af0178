#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reference_real.py::test_training_loop_of_the_real_classes_hooks_on_equals_hooks_off tests/test_gpu_rccl.py -q > $O/tests.txt 2>&1; tail -25 $O/tests.txt | cut -c1-3000
for o in random morton random morton; do timeout 200 python bench.py --cpu-sample 0 --steps 60 --warmup 10 --order $o 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_step']['kernel_ms_per_step']
print('order=$o', round(d['ms_per_step'], 4), ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"; done > $O/ab_order.txt 2>&1; cat $O/ab_order.txt
