#!/bin/bash
# The round's artifact session (one gpurun call): full -m gpu suite, the default bench line, rocprofv3 kernel stats of the same
# command per precision, the PMC passes (separate --pmc runs), kernel stats + PMC of the fp32-class training step.
#   bash tools/gpu_round_session.sh TAG      -> everything under gpurun_out/TAG/ ; copy what is to be judged into profiles/
cd "${GRAFT_REPO_ROOT:-/root/repo}"; TAG=${1:-round}; O=gpurun_out/$TAG; mkdir -p $O; R=$PWD; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("value %.0f frac %.3f avg_launch %.2f ms clk %.3f busy %.3f traffic %.1f GB eager x%.2f (unchunked x%.2f)" % (d["value"], r["frac"], r["avg_launch_ms"], r.get("shader_clock_ghz_during_kernel",0), r.get("mfma_busy_frac_measured_in_run",0), (r.get("traffic") or 0)/1e9, d.get("speedup_vs_torch_eager_gpu",0), d.get("speedup_vs_torch_eager_gpu_unchunked_16384",0)))
p=d.get("f16_path"); print("f16 peer %.0f frac %.3f" % (p["value"], p["roofline"]["frac"])) if p else None
for k,v in d.get("extra",{}).get("configs",{}).items(): print(k, json.dumps(v)[:260])
print({k:d[k] for k in ("latency_4096_rays_ms","encode_ms","cpu_baseline","psnr_db") if k in d})
PY
for prec in f16x3 f16; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_$prec -o st -- python $R/bench.py --prec $prec --steps 20 --warmup 5 --no-peer --no-latency --no-cpu-baseline --no-eager-baseline --no-f32-check --no-extras --no-live-pmc > $R/$O/stats_$prec.log 2>&1)
  f=$(find $O/stats_$prec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-200
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats_train -o st -- python $R/tools/gpu_train_f16x3_quick.py > $R/$O/stats_train.log 2>&1); grep "ms/step" $O/stats_train.log
f=$(find $O/stats_train -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200
PREC=f16x3 bash tools/collect_pmc.sh > $O/pmc_f16x3.log 2>&1; tail -25 $O/pmc_f16x3.log; cp -r gpurun_out/pmc_f16x3 $O/ 2>/dev/null
PREC=f16 bash tools/collect_pmc.sh > $O/pmc_f16.log 2>&1; tail -12 $O/pmc_f16.log; cp -r gpurun_out/pmc_f16 $O/ 2>/dev/null
bash tools/collect_pmc_train_f16x3.sh > $O/pmc_train_f16x3.txt 2>&1; tail -12 $O/pmc_train_f16x3.txt
