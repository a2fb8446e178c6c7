#!/bin/bash
# Copy the judged artifacts of one artifact session (tools/gpu_round_session.sh TAG -> gpurun_out/TAG/) into profiles/ under the
# round's prefix:   bash tools/publish_session.sh TAG r06
# (run HERE after the gpurun call came back; gpurun_out/ is scratch, profiles/ is tracked)
set -e
cd "$(dirname "$0")/.."; TAG=$1; R=$2; S=gpurun_out/$TAG; P=profiles
[ -d "$S" ] || { echo "no $S"; exit 1; }
grep '^{' $S/bench.json | tail -1 > $P/${R}_bench_f16x3.json
cp $S/stats_f16x3/st_kernel_stats.csv $P/${R}_bench_f16x3_kernel_stats.csv
cp $S/stats_f16/st_kernel_stats.csv $P/${R}_bench_f16_kernel_stats.csv
cp $S/stats_train/st_kernel_stats.csv $P/${R}_train_step_f16x3_kernel_stats.csv
cp $S/pmc_f16x3/pmc_eval_split_kernel.json $P/${R}_bench_f16x3_pmc_eval_split_kernel.json
cp $S/pmc_f16/pmc_eval_kernel.json $P/${R}_bench_f16_pmc_eval_kernel.json
cp $S/pmc_train_f16x3.txt $P/${R}_train_step_f16x3_pmc.txt
{ echo "# tail of 'python -m pytest tests -x -q -m gpu' in session $TAG (sources: see _kernel_source_sha16 / head in ${R}_bench_f16x3.json)"; tail -4 $S/pytest.log; } > $P/${R}_gpu_tests.txt
grep "ms/step" $S/stats_train.log > $P/${R}_train_step_f16x3_ms.txt || true
python - $P/${R}_bench_f16x3.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print("published: %.0f rays/s, frac %.3f, avg launch %.2f ms, clock %.3f GHz, head %s" % (
    d["value"], r["frac"], r["avg_launch_ms"], r.get("shader_clock_ghz_during_kernel", 0), d.get("head", "?")))
PY
