#!/bin/bash
# Training-step timing + per-kernel table on the GPU box (through gpurun): tools/gpu_train_bench.py --quick alone, then the
# same command under rocprofv3 --kernel-trace --stats (csv).  Outputs gpurun_out/train_quick.txt, gpurun_out/train_kernels.txt
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
timeout 300 python $REPO/tools/gpu_train_bench.py --quick > $REPO/gpurun_out/train_quick.txt 2>&1
rm -rf $REPO/gpurun_out/prof_train
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_train -o tr -- python $REPO/tools/gpu_train_bench.py --quick > $REPO/gpurun_out/train_prof.log 2>&1
python $REPO/tools/kernel_stats_summary.py $(find $REPO/gpurun_out/prof_train -name "*kernel_stats.csv" | head -1) 28 > $REPO/gpurun_out/train_kernels.txt 2>&1
tail -3 $REPO/gpurun_out/train_quick.txt
head -14 $REPO/gpurun_out/train_kernels.txt
