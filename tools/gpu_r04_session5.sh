#!/bin/bash
# round 4, GPU session 5: the whole suite + smoke on the final sources
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s5; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -q -m gpu -s > $OUT/pytest_all.log 2>&1; echo "pytest(all) rc=$?" | tee -a $OUT/pytest_all.log; grep -v amdgpu.ids $OUT/pytest_all.log | grep -i "passed\|failed\|error\|trained\|view maximum\|bin with" | tail -40
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee $OUT/smoke.log
