#!/bin/bash
# PMC counters of the FUSED fp32-class (f16x3) training-step kernels (tools/gpu_train_f16x3_quick.py): one rocprofv3 --pmc pass per counter group,
# no trace domains besides --kernel-trace; per-kernel averages printed by the inline summary.  Usage (through gpurun):
#   bash tools/collect_pmc_train_f16x3.sh
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_train_f16x3"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export PIXELNERF_SATURATION_GUARD=off
CMD="python $REPO/tools/gpu_train_f16x3_quick.py"
run() { local name=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "pmc_$name" -- $CMD > "$OUT/$name.log" 2>&1; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
python - "$OUT" "$REPO" <<'PY'
import csv, glob, hashlib, os, sys
out = sys.argv[1]
# identity of the kernels these counters were collected on: hash of the training-step units + shared headers
h = hashlib.sha256()
for f in ("pnr_bwd.hip", "pnr_pack.hip", "pnr_split.hip", "pnr_device.h", "pnr_layout.h"):
    h.update(open(os.path.join(sys.argv[2], "pixel-nerf_amd", "csrc", f), "rb").read())
print("# _kernel_source_sha16 %s  (sha256 of pnr_bwd.hip + pnr_pack.hip + pnr_split.hip + pnr_device.h + pnr_layout.h, first 16 hex digits)" % h.hexdigest()[:16])
acc = {}
for path in sorted(glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if not any(s in k for s in ("dw_split_kernel", "dw_split_wide_kernel", "bwd_split_kernel", "eval_split_kernel", "latent_scatter", "fold_kernel", "fold_split", "gemm3", "lin_out_grad_f32")):
            continue
        name = k.split("(")[0].replace("void pnr::", "")[:60]
        per.setdefault((name, row["Counter_Name"]), {}).setdefault(row["Dispatch_Id"], 0.0)
        per[(name, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for (name, ctr), d in per.items():
        acc.setdefault(name, {})[ctr] = sum(d.values()) / len(d)
for name, c in sorted(acc.items()):
    line = [name]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and c.get("SQ_BUSY_CU_CYCLES"):
        line.append("MFMA busy %.1f %%" % (100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_BUSY_CU_CYCLES"])))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        line.append("LDS bank conflicts %.1f %% of LDS-active cycles" % (100 * c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
    if c.get("SQ_WAVE_CYCLES"):
        line.append("waiting %.0f %% of wave cycles" % (100 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]))
    if "FETCH_SIZE" in c:
        line.append("fabric reads %.0f MB/launch" % (2 * c["FETCH_SIZE"] * 1024 / 1e6))
    if "WRITE_SIZE" in c:
        line.append("writes %.0f MB/launch" % (c["WRITE_SIZE"] * 1024 / 1e6))
    print("  ".join(line))
PY
