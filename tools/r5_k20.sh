#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c11; mkdir -p $O
for a in "--steps 20 --warmup 5" "--steps 100 --warmup 30" "--steps 20 --warmup 5" "--steps 100 --warmup 30" "--steps 20 --warmup 5 --subpixel-offset none" "--steps 20 --warmup 5"; do
timeout 300 python bench.py --cpu-sample 0 $a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_step']['kernel_ms_per_step']
print('[$a]', round(d['ms_per_step'], 4), d['roofline']['avg_launch_ms'], ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
done > $O/k20.txt 2>&1; cat $O/k20.txt
