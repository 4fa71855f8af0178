#!/bin/bash
# A/B on ONE box: alternates bench.py runs between prebuilt libraries (SFGS_LIB), ROUNDS times (default 3).
# usage: [ROUNDS=n] tools/ab.sh libA.so libB.so ... [-- bench args]
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for r in $(seq 1 ${ROUNDS:-3}); do for l in "${LIBS[@]}"; do
  SFGS_LIB=$PWD/$l timeout 200 python bench.py --cpu-sample 0 "$@" 2>/dev/null < /dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_step']['kernel_ms_per_step']
print('$l', round(d['ms_per_step'], 4), ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
done; done
