#!/bin/bash
# Collect the PMC counters of the fused network kernel for `bench.py` on the GPU box: ONE rocprofv3 --pmc pass per
# counter group (no trace domains besides --kernel-trace), outputs under gpurun_out/pmc_<prec>/, summary JSON printed by
# tools/pmc_summarize.py.  Usage (through gpurun):  PREC=f16x3|f16 bash tools/collect_pmc.sh [extra bench.py flags]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
PREC="${PREC:-f16x3}"
case "$PREC" in f16x3) KERNEL=eval_split_kernel ;; *) KERNEL=eval_kernel ;; esac
OUT="$REPO/gpurun_out/pmc_$PREC"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PIXELNERF_SATURATION_GUARD=off  # every launch of the profiled process is the plain instantiation (the guard runs once per new weights otherwise)
CMD="python $REPO/bench.py --prec $PREC --steps 2 --warmup 1 --no-peer --no-latency --no-cpu-baseline --no-eager-baseline --no-f32-check --no-extras --no-live-pmc $*"
run() {  # name, counters...
    local name=$1; shift
    timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "pmc_$name" -- $CMD > "$OUT/$name.log" 2>&1
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
run sq SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run tcc TCC_HIT_sum TCC_MISS_sum
python "$REPO/tools/pmc_summarize.py" "$OUT" "$CMD" "$KERNEL" "$PREC"
