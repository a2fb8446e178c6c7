#!/bin/bash
# rocprofv3 kernel stats of the composed path's forward + backward (tools/gpu_composed_bench.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/cp
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/cp/st -o st -- python $R/tools/gpu_composed_bench.py > $R/gpurun_out/cp/run.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/cp/run.log | tail -5
f=$(find $R/gpurun_out/cp/st -name "*kernel_stats.csv" | head -1); python $R/tools/kernel_stats_summary.py "$f" | head -40 | cut -c1-200
