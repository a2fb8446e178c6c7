#!/bin/bash
# PMC pass over the fp32-class (f16x3) training step: where the split-GEMM kernels' wave cycles go
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/f
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/f/pmc_$name -o p -- python -c "
import sys, torch; sys.path.insert(0, '$R')
import bench
r = bench.extra_train_step(torch.device('cuda:0'), 'f16x3', steps=3, warmup=1, with_graph=False)
" > $R/gpurun_out/f/pmc_$name.log 2>&1; }
run a SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run c SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob("$R/gpurun_out/f/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "gemm3" not in k: continue
        k = k[k.index("gemm3"):k.index("(")]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])].add(row["Dispatch_Id"])
for k, d in acc.items():
    n = {c: len(cnt[(k, c)]) for c in d}
    print(k, {c: "%.3g" % (v / n[c]) for c, v in d.items()})
    g = lambda c: d[c] / n[c] if c in d else float("nan")
    print("   mfma busy %.3f  valu-active/wave-cycles %.3f  lds-active %.3f  wait_inst %.3f  wait_any %.3f  bank conflict %.3f" % (
        g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")), g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_LDS") / g("SQ_WAVE_CYCLES"),
        g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
PY
