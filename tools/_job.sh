export TMPDIR=/tmp; O=gpurun_out/c16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_robustness.py tests/test_gpu_joint_render.py tests/test_gpu_fullsize_parity.py tests/test_gpu_fullsize.py tests/test_gpu_launch_hints.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests.txt
timeout 300 python tools/bitcompare.py > $O/bits_main.json 2>/dev/null
for i in 1 2 3; do timeout 300 python bench.py --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 4), d['train_shaped']['ms_per_step'], d['box']['hbm_tbs'], d['box']['valu_tflops'], d['roofline_step']['kernel_ms_per_step'])"; done | tee $O/bench3.txt
timeout 300 python bench.py --forward-only --cpu-sample 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', round(d['value'], 1), d['ms_per_step'])" | tee -a $O/bench3.txt
