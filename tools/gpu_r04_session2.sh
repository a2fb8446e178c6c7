#!/bin/bash
# round 4, GPU session 2: debug of the two failing tests, op profile of the fp32-class training step, its ms/step (new fold +
# lin_out reduction) against the fp32-MFMA-fold twin, then the whole suite without -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s2; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/gpu_debug_generic.py 2>&1 | grep -v amdgpu.ids | tee $OUT/debug.log
echo "=== train step f16x3 (default build)"; timeout 300 python tools/gpu_train_f16x3_quick.py 2>&1 | grep -v amdgpu.ids | tee $OUT/train_default.log
echo "=== train step f16x3 (fp32-MFMA fold twin)"; PIXELNERF_ALLOW_VARIANT=1 PIXELNERF_HIP_LIB=$PWD/build/libpnr_t_foldf32.so timeout 300 python tools/gpu_train_f16x3_quick.py 2>&1 | grep -v amdgpu.ids | tee $OUT/train_foldf32.log
timeout 300 python tools/gpu_train_opprofile.py f16x3 2>&1 | grep -v amdgpu.ids | head -60 | tee $OUT/opprofile.log
bash tools/gpu_train_f16x3_prof.sh f16x3 2>&1 | tee $OUT/train_stats.log; cp gpurun_out/f/st/*/*kernel_stats.csv $OUT/train_kernel_stats.csv 2>/dev/null || find gpurun_out/f/st -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats.csv \;
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_all.log; grep -v amdgpu.ids $OUT/pytest_all.log | tail -40
