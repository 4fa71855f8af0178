#!/bin/bash
# A/B on ONE box between ENVIRONMENT settings of the shipped library (alternating processes), ROUNDS times (default 3).
# usage: [ROUNDS=n] tools/ab_env.sh "NAME1=VAL ..." "NAME2=VAL ..." [-- bench args]      ("" = no extra environment)
SETS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do SETS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for r in $(seq 1 ${ROUNDS:-3}); do for e in "${SETS[@]}"; do
  env $e timeout 200 python bench.py --cpu-sample 0 "$@" 2>/dev/null < /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_step']['kernel_ms_per_step']
print('[$e]', round(d['ms_per_step'], 4), ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
done; done
