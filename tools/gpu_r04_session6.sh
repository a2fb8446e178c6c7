#!/bin/bash
# round 4, GPU session 6: whole suite on the composed-path sources (+ merged / vectorised pack kernel, lazy content fingerprint),
# then the fp32-class training step: ms/step + kernel stats of the default build and of two TIMING-ONLY twins of dw_split_kernel
# (no global loads after the first slab / no MFMAs) that say which side of that kernel the time is on
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
OUT=gpurun_out/r04_s6; mkdir -p $OUT gpurun_out/f
export TMPDIR=/tmp
timeout 1500 python -m pytest ${PYTEST_ARGS:-tests} -q -m gpu -s > $OUT/pytest_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_all.log
grep -v amdgpu.ids $OUT/pytest_all.log | grep -i "passed\|failed\|error\|ResnetFC \|variant \|renderer around" | tail -40
grep -v amdgpu.ids $OUT/pytest_all.log | grep -B5 -A40 "^___\|Error" | head -150
[ -n "$SKIP_QUICK" ] || timeout 300 python tools/gpu_train_f16x3_quick.py 2>&1 | grep -v amdgpu.ids | tee $OUT/train_default.log
prof() {  # name, lib
    rm -rf $R/gpurun_out/f/st_$1
    ( cd /tmp; if [ -n "$2" ]; then export PIXELNERF_ALLOW_VARIANT=1 PIXELNERF_HIP_LIB=$2; fi
      timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f/st_$1 -o st -- python -c "
import sys, torch; sys.path.insert(0, '$R')
import bench
r = bench.extra_train_step(torch.device('cuda:0'), 'f16x3', steps=6, warmup=2, with_graph=False)
print(r['ms_per_step'])
" > $R/gpurun_out/f/run_$1.log 2>&1 )
    f=$(find $R/gpurun_out/f/st_$1 -name "*kernel_stats.csv" | head -1)
    cp "$f" $OUT/train_stats_$1.csv
    echo "--- $1: $(tail -1 $R/gpurun_out/f/run_$1.log) ms/step under rocprof"
    python tools/kernel_stats_summary.py "$f" 8 | head -24 | cut -c1-170
}
prof default ""
prof dw_noload $R/build/libpnr_dw_noload.so
prof dw_nomfma $R/build/libpnr_dw_nomfma.so
