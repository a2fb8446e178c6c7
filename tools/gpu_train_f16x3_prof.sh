#!/bin/bash
# rocprofv3 kernel stats of the fp32-class (f16x3) training step: which of the split-GEMM launches the time goes to
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/f
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f/st -o st -- python -c "
import sys, torch; sys.path.insert(0, '$R')
import bench
r = bench.extra_train_step(torch.device('cuda:0'), '${1:-f16x3}', steps=6, warmup=2, with_graph=False)
print(r['ms_per_step'])
" > $R/gpurun_out/f/run.log 2>&1
tail -2 $R/gpurun_out/f/run.log
f=$(find $R/gpurun_out/f/st -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-220
