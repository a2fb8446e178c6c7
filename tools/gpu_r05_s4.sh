cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r05_s4; O=gpurun_out/r05_s4
python tools/gpu_train_opprofile.py f16x3 > $O/opprofile.txt 2>&1; head -70 $O/opprofile.txt
timeout 1700 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
