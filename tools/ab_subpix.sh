for r in 1 2 3; do for m in none zeros; do
  timeout 200 python bench.py --cpu-sample 0 --subpixel-offset $m 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline_step']['kernel_ms_per_step']
print('$m', round(d['ms_per_step'], 4), ' '.join(f'{n}={v:.4f}' for n, v in k.items()))"
done; done
