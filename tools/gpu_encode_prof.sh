#!/bin/bash
# rocprofv3 kernel durations of the encoder output formatting (tools/gpu_encode_bench.py): the wall-clock figures of that tool
# include the host side of a ~0.1 ms call
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/enc
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/enc/st -o st -- python $R/tools/gpu_encode_bench.py > $R/gpurun_out/enc/run.log 2>&1
grep -v amdgpu.ids $R/gpurun_out/enc/run.log
f=$(find $R/gpurun_out/enc/st -name "*kernel_stats.csv" | head -1); head -8 "$f" | cut -c1-200
