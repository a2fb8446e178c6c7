#!/bin/bash
# Headline-only same-box A/B of variant libraries:  bash tools/gpu_ab_sn64.sh TAG libA libB [libC ...]   (two alternating rounds)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PIXELNERF_ALLOW_VARIANT=1; R=$PWD; TAG=$1; shift; mkdir -p gpurun_out/$TAG
for i in 1 2; do for v in "$@"; do
PIXELNERF_HIP_LIB=$R/build/libpnr_$v.so python bench.py --prec f16x3 --steps 10 --warmup 3 --no-peer --no-extras --no-cpu-baseline --no-eager-baseline --no-latency --no-live-pmc --no-f32-check 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$v $i sn64 %.0f rays/s frac %.3f' % (d['value'], d['roofline']['frac']))" | tee -a gpurun_out/$TAG/ab.txt
done; done
