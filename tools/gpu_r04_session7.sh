#!/bin/bash
# round 4, GPU session 7: training copy-outs riding inside the GEMM loops (fp32-class forward + backward chain): whole suite,
# then ms/step of the fp32-class training step against the copy-out-in-front twin (-DPNR_X_DUMP_FRONT), same box, + kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
OUT=gpurun_out/r04_s7; mkdir -p $OUT gpurun_out/f
export TMPDIR=/tmp
timeout 1500 python -m pytest ${PYTEST_ARGS:-tests} -q -m gpu -s > $OUT/pytest_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_all.log
grep -v amdgpu.ids $OUT/pytest_all.log | grep -E "^(FAILED|ERROR)|passed|failed" | tail -20
grep -v amdgpu.ids $OUT/pytest_all.log | grep -B5 -A30 "^___" | head -120
for rep in 1 2; do
echo "=== train step f16x3 (default build)"; timeout 300 python tools/gpu_train_f16x3_quick.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/train_default.log
echo "=== train step f16x3 (copy-out in front of the GEMMs twin)"; PIXELNERF_ALLOW_VARIANT=1 PIXELNERF_HIP_LIB=$R/build/libpnr_dump_front.so timeout 300 python tools/gpu_train_f16x3_quick.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/train_dump_front.log
done
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
    echo "--- $1"
    python tools/kernel_stats_summary.py "$f" 8 | head -12 | cut -c1-170
}
prof default ""
prof dump_front $R/build/libpnr_dump_front.so
