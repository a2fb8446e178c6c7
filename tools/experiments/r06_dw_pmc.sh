export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r06_s48; mkdir -p $O; cd /tmp
for form in lds wide; do
  PNR_DW_FORM=$form timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_$form -o p -- python $R/tools/gpu_dw_bench.py 32768 > $O/$form.log 2>&1
  PNR_DW_FORM=$form timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/pmc2_$form -o p -- python $R/tools/gpu_dw_bench.py 32768 > $O/${form}2.log 2>&1
done
cd $R
python - $O <<'PY'
import csv, glob, sys, os
for d in sorted(glob.glob(os.path.join(sys.argv[1], "pmc*"))):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if "dw_split" not in row["Kernel_Name"]: continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print(os.path.basename(d), {k: "%.3g" % (sum(v) / len(v)) for k, v in acc.items()}, "n", {k: len(v) for k, v in acc.items()})
PY
