cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05_s1; O=gpurun_out/r05_s1
./tools/ubench/split_gemm.bin 5 > $O/split_gemm.txt 2>&1; cat $O/split_gemm.txt
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
r=d["roofline"]; print("value %.0f frac %.3f avg_launch %.2f ms clk %.3f busy %.3f eager x%.2f" % (d["value"], r["frac"], r["avg_launch_ms"], r.get("shader_clock_ghz_during_kernel",0), r.get("mfma_busy_frac_measured_in_run",0), d.get("speedup_vs_torch_eager_gpu",0)))
for k,v in d.get("extra",{}).get("configs",{}).items(): print(k, json.dumps(v)[:300])
PY
