#!/bin/bash
# One gpurun session: [tests] + bench line + rocprofv3 kernel stats + PMC passes, everything under gpurun_out/<tag>/.
#   bash tools/gpu_session.sh TAG "pytest args or empty" [ab] [pmc]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${1:-s}; TESTS=${2:-}; shift; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then
    timeout 1500 python -m pytest $TESTS -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
fi
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python - <<PY
import json
try:
    d=json.loads([l for l in open("$OUT/bench.json") if l.startswith("{")][-1])
    r=d["roofline"]; print("value %.0f rays/s dtype %s frac %.3f exec %.0f TF avg_launch %.2f ms psnr %s eager x%.2f" % (d["value"], d["dtype"], r["frac"], r["executed_mfma_tflops"], r["avg_launch_ms"], d.get("psnr_db"), d.get("speedup_vs_torch_eager_gpu", 0)))
    p=d.get("f16_path")
    if p: print("f16 peer %.0f rays/s frac %.3f" % (p["value"], p["roofline"]["frac"]))
    for k,v in d.get("extra",{}).get("configs",{}).items(): print(k, json.dumps(v)[:400])
    print({k:d[k] for k in ("latency_4096_rays_ms","encode_ms","cpu_baseline","cpu_baseline_config1") if k in d})
except Exception as e: print("bench parse failed", e)
PY
for what in "$@"; do
case $what in
ab) bash tools/gpu_split_ab.sh ;;
stats)
    for prec in f16x3 f16; do
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/stats_$prec -o st -- python $OLDPWD/bench.py --prec $prec --steps 20 --warmup 5 --no-peer --no-latency --no-cpu-baseline --no-eager-baseline --no-f32-check --no-extras --no-live-pmc > $OLDPWD/$OUT/stats_$prec.log 2>&1)
        f=$(find $OUT/stats_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
    done ;;
pmc)
    PREC=f16x3 bash tools/collect_pmc.sh > $OUT/pmc_f16x3.log 2>&1; tail -30 $OUT/pmc_f16x3.log
    PREC=f16 bash tools/collect_pmc.sh > $OUT/pmc_f16.log 2>&1; tail -30 $OUT/pmc_f16.log ;;
esac
done
