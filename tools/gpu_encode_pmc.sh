#!/bin/bash
# PMC passes over the encoder output formatting kernel (tools/gpu_encode_bench.py): where its wave cycles go
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD; mkdir -p gpurun_out/enc
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/enc/pmc_$name -o p -- python $R/tools/gpu_encode_bench.py > $R/gpurun_out/enc/pmc_$name.log 2>&1; }
run a SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
run c SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU
run d SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
run e SQ_WAVES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/enc/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "pyramid" not in k and "nchw_to_nhwc" not in k: continue
        acc[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v); print("   %-26s n=%d min %.4g median %.4g max %.4g" % (c, len(v), v[0], v[len(v)//2], v[-1]))
PY
