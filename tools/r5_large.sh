#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c12; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reference_real.py -q -k "training_viewport" > $O/tests.txt 2>&1; tail -15 $O/tests.txt | cut -c1-1500
cat gpurun_out/reference_real_render_1080p.jsonl 2>/dev/null | cut -c1-500
