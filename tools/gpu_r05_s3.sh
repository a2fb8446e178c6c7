cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05_s3; O=gpurun_out/r05_s3
B="python bench.py --prec f16x3 --steps 10 --warmup 3 --no-peer --no-extras --no-cpu-baseline --no-eager-baseline --no-latency --no-live-pmc --no-f32-check"
show() { python - "$1" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print(sys.argv[1], "value %.0f frac %.3f avg_launch %.2f ms psnr %s" % (d["value"], r["frac"], r["avg_launch_ms"], d.get("psnr_db")))
PY
}
$B > $O/bench_prod_1.json 2> $O/bench_prod_1.err; show $O/bench_prod_1.json
export PIXELNERF_HIP_LIB=build/libpnr_v4.so PIXELNERF_ALLOW_VARIANT=1
timeout 600 python -m pytest tests/test_hip_split.py -x -q -m gpu -k "not f16-" > $O/pytest_v4.log 2>&1; echo "pytest v4 rc=$?"; tail -15 $O/pytest_v4.log
$B > $O/bench_v4_1.json 2> $O/bench_v4_1.err; show $O/bench_v4_1.json; tail -3 $O/bench_v4_1.err
unset PIXELNERF_HIP_LIB PIXELNERF_ALLOW_VARIANT
$B > $O/bench_prod_2.json 2> $O/bench_prod_2.err; show $O/bench_prod_2.json
export PIXELNERF_HIP_LIB=build/libpnr_v4.so PIXELNERF_ALLOW_VARIANT=1
$B > $O/bench_v4_2.json 2> $O/bench_v4_2.err; show $O/bench_v4_2.json
unset PIXELNERF_HIP_LIB PIXELNERF_ALLOW_VARIANT
python tools/gpu_train_opprofile.py f16x3 > $O/opprofile.txt 2>&1; head -60 $O/opprofile.txt
