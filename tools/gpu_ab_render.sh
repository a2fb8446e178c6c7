#!/bin/bash
# Same-box A/B of variant libraries on the three render workloads:  bash tools/gpu_ab_render.sh TAG libA libB [rounds]
#   PREC=f16 selects the opt-in 16-bit kernel (default f16x3).
#   (libraries from tools/build_objs.sh / build_variant.sh: build/libpnr_<name>.so; alternating runs; sn64 headline via bench.py,
#   srn_car / DTU via bench.extra_render_config)
PREC=${PREC:-f16x3}
cd "${GRAFT_REPO_ROOT:-/root/repo}"; TAG=$1; A=$2; B=$3; N=${4:-2}; O=gpurun_out/$TAG; mkdir -p $O; R=$PWD
export PIXELNERF_ALLOW_VARIANT=1
for i in $(seq 1 $N); do for v in $A $B; do
PIXELNERF_HIP_LIB=$R/build/libpnr_$v.so python -c "
import json, subprocess, sys, torch, bench
dev = torch.device('cuda:0')
out = subprocess.run([sys.executable, 'bench.py', '--prec', '$PREC', '--steps', '10', '--warmup', '3', '--no-peer', '--no-extras', '--no-cpu-baseline',
                      '--no-eager-baseline', '--no-latency', '--no-live-pmc', '--no-f32-check'], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith('{')][-1])
print('$v $i sn64 %.0f rays/s frac %.3f' % (d['value'], d['roofline']['frac']))
for name, n in (('srn_car', 4), ('dtu', 1)):
    r = bench.extra_render_config(dev, name, n, n_oracle=16, n_f32=1024, steps=3)
    print('$v $i', name, '%.0f rays/s' % r['$PREC']['rays_per_s'], 'psnr vs oracle %.1f' % r['$PREC']['psnr_db_vs_cpu_oracle'])
" 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done; done
