#!/bin/bash
# round 4, GPU session 4: the whole suite, the bench line, rocprofv3 kernel stats of both precisions, PMC passes (inference + training)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s4; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_all.log; grep -v amdgpu.ids $OUT/pytest_all.log | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
    r=d["roofline"]; print("value %.0f rays/s dtype %s frac %.3f exec %.0f TF avg_launch %.2f ms psnr %s eager x%.2f (unchunked x%.2f)" % (d["value"], d["dtype"], r["frac"], r["executed_mfma_tflops"], r["avg_launch_ms"], d.get("psnr_db"), d.get("speedup_vs_torch_eager_gpu", 0), d.get("speedup_vs_torch_eager_gpu_unchunked_16384", 0)))
    p=d.get("f16_path")
    if p: print("f16 peer %.0f rays/s frac %.3f" % (p["value"], p["roofline"]["frac"]))
    for k in ("torch_eager_gpu_baseline","torch_eager_gpu_baseline_unchunked_16384","cpu_baseline","latency_4096_rays_ms"): print(k, json.dumps(d.get(k))[:300])
    for k,v in d.get("extra",{}).get("configs",{}).items(): print(k, json.dumps(v)[:330])
except Exception as e: print("bench parse failed", e); print(open("$OUT/bench.err").read()[-2000:])
PY
export PIXELNERF_SATURATION_GUARD=off
for prec in f16x3 f16; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stats_$prec -o st -- python $OLDPWD/bench.py --prec $prec --steps 20 --warmup 5 --no-peer --no-latency --no-cpu-baseline --no-eager-baseline --no-f32-check --no-extras --no-live-pmc > $OLDPWD/$OUT/stats_$prec.log 2>&1)
    f=$(find $OUT/stats_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bench_${prec}_kernel_stats.csv && head -6 "$f" | cut -c1-200
    rm -rf $OUT/stats_$prec
done
PREC=f16x3 bash tools/collect_pmc.sh > $OUT/pmc_f16x3.log 2>&1; tail -25 $OUT/pmc_f16x3.log; cp gpurun_out/pmc_f16x3/pmc_eval_split_kernel.json $OUT/ 2>/dev/null
PREC=f16 bash tools/collect_pmc.sh > $OUT/pmc_f16.log 2>&1; tail -25 $OUT/pmc_f16.log; cp gpurun_out/pmc_f16/pmc_eval_kernel.json $OUT/ 2>/dev/null
bash tools/gpu_train_f16x3_prof.sh f16x3 2>&1 | tail -14; find gpurun_out/f/st -name "*kernel_stats.csv" -exec cp {} $OUT/train_f16x3_kernel_stats.csv \;
bash tools/collect_pmc_train_f16x3.sh > $OUT/pmc_train_f16x3.txt 2>&1; tail -12 $OUT/pmc_train_f16x3.txt
rm -rf gpurun_out/pmc_f16x3 gpurun_out/pmc_f16 gpurun_out/pmc_train_f16x3 gpurun_out/f
