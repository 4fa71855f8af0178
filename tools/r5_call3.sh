#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r5c3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_reference_real.py::test_training_loop_of_the_real_classes_hooks_on_equals_hooks_off tests/test_gpu_joint_fullsize.py tests/test_gpu_densify_masks_fullsize.py tests/test_gpu_robustness.py "tests/test_gpu_raster.py::test_forward_backward_parity[configs0_50k_800sq]" -q -s > $O/tests.txt 2>&1; tail -40 $O/tests.txt | cut -c1-2500
