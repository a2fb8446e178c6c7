#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc CSVs written by tools/collect_pmc.sh to one JSON: per counter the values of every
launch of the fused network kernel (pnr::eval_kernel, or pnr::eval_split_kernel for f16x3) and their average, plus derived figures.  FETCH_SIZE and
WRITE_SIZE are in KiB; per MI355X_MICROARCH.md FETCH_SIZE is doubled on gfx950."""
import csv
import glob
import json
import os
import sys


def main():
    out_dir, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    kname = sys.argv[3] if len(sys.argv) > 3 else "eval_kernel"  # substring of the kernel name
    prec = sys.argv[4] if len(sys.argv) > 4 else "f16"
    res = {}
    kern = {}
    for path in sorted(glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True)):
        per = {}
        for row in csv.DictReader(open(path)):
            if kname + "<" not in row["Kernel_Name"] and kname + "I" not in row["Kernel_Name"]:
                continue
            per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            kern = {"VGPR": row["VGPR_Count"], "AGPR": row["Accum_VGPR_Count"], "SGPR": row["SGPR_Count"],
                    "LDS": row["LDS_Block_Size"], "Scratch": row["Scratch_Size"], "grid": row["Grid_Size"], "wg": row["Workgroup_Size"],
                    "name": row["Kernel_Name"][:120]}
        for name, d in per.items():
            vals = [d[k] for k in sorted(d, key=int)]
            res[name] = {"avg_per_launch": sum(vals) / len(vals), "launches": len(vals), "values": vals}
    g = lambda k: res[k]["avg_per_launch"] if k in res else None  # noqa: E731
    der = {}
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        der["L2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
    if g("SQ_VALU_MFMA_BUSY_CYCLES") is not None and g("SQ_BUSY_CU_CYCLES") is not None:
        der["mfma_busy_frac"] = g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES"))  # 4 SIMDs per CU
    if g("GRBM_GUI_ACTIVE") is not None:
        der["gpu_active_cycles_per_launch"] = g("GRBM_GUI_ACTIVE")  # / kernel duration = the average shader clock
    if g("FETCH_SIZE") is not None:
        der["fabric_read_bytes_per_launch"] = 2.0 * g("FETCH_SIZE") * 1024.0
    if g("WRITE_SIZE") is not None:
        der["hbm_write_bytes_per_launch"] = g("WRITE_SIZE") * 1024.0
    if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_LDS_IDX_ACTIVE"):
        der["lds_bank_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
    res["_kernel"], res["_derived"] = kern, der
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["_kernel_source_sha16"] = bench.kernel_source_sha16(prec)  # bench.py refuses this profile once the kernel sources change
    res["_note"] = ("fused network kernel launches of `%s`; one rocprofv3 --pmc pass per counter group; FETCH_SIZE/WRITE_SIZE in KiB, "
                    "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)" % cmd)
    json.dump(res, open(os.path.join(out_dir, "pmc_%s.json" % kname), "w"), indent=1)
    print(json.dumps({"derived": der, "kernel": kern, "counters": {k: v["avg_per_launch"] for k, v in res.items() if not k.startswith("_")}}, indent=1))


if __name__ == "__main__":
    main()
