#!/bin/bash
# round 4, GPU session 1: new-kernel parity first, then same-box A/B of the own-K-first block against its twins, then the whole suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s1; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_split.py tests/test_hip_fullsize.py tests/test_hip_backward_f32.py tests/test_hip_trained_weights.py \
    tests/test_hip_generic_training.py tests/test_hip_features.py -x -q -m gpu -s > $OUT/pytest_new.log 2>&1
echo "pytest(new) rc=$?" | tee -a $OUT/pytest_new.log; grep -v "amdgpu.ids" $OUT/pytest_new.log | tail -40
bash tools/gpu_split_ab.sh 2>&1 | tail -40
cp gpurun_out/sab_*.txt $OUT/ 2>/dev/null
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_all.log; tail -15 $OUT/pytest_all.log
