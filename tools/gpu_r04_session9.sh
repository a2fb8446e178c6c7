#!/bin/bash
# round 4, GPU session 9: feature-phase work of the fp32-class kernel (corner-row reuse in the table gather, lin_in operand
# padding written once, packed 32-bit LDS stores of the positional code): parity tests of the kernel, then same-box A/B against
# the twins (-DPNR_X_GATHER_NOREUSE, -DPNR_X_GEOM_PAD_ALWAYS, both) with phase tables
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r04_s9; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_hip_split.py tests/test_hip_parity.py tests/test_hip_features.py tests/test_hip_adversarial.py tests/test_api_gpu.py -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -v amdgpu.ids $OUT/pytest.log | grep -E "^(FAILED|ERROR)|passed|failed" | tail
bash tools/gpu_split_ab.sh
cat gpurun_out/sab_default.txt gpurun_out/sab_s_noreuse.txt gpurun_out/sab_s_padalways.txt gpurun_out/sab_s_old_nt.txt > $OUT/ab.txt
grep -h "===\|rays/s\|geometry\|gather \|table " $OUT/ab.txt | cut -c1-200
