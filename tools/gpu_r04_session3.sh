#!/bin/bash
# round 4, GPU session 3: encoder-graph debug, the tests that failed in session 2, then the whole suite, then the train step
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s3; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/gpu_debug_encoder.py 2>&1 | grep -v amdgpu.ids | tee $OUT/debug_encoder.log
timeout 300 python tools/gpu_debug_generic.py 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/debug_generic.log
timeout 1200 python -m pytest tests/test_hip_split.py tests/test_hip_generic_training.py tests/test_hip_trained_weights.py tests/test_hip_stage_sweep.py \
    tests/test_api_gpu.py -q -m gpu -s > $OUT/pytest_sel.log 2>&1
echo "pytest(sel) rc=$?" | tee -a $OUT/pytest_sel.log; grep -v "amdgpu.ids" $OUT/pytest_sel.log | grep -i "trained\|guarded\|generic\|passed\|failed\|error" | tail -40
echo "=== train step f16x3"; timeout 300 python tools/gpu_train_f16x3_quick.py 2>&1 | grep -v amdgpu.ids | tee $OUT/train_default.log
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_all.log; grep -v amdgpu.ids $OUT/pytest_all.log | tail -15
