#!/bin/bash
# A/B of library variants of the fp32-class kernel on the GPU box: every build/libpnr_s_*.so plus the default library runs
# the f16x3 quick bench (sn64 + multi-view shapes) and, unless the name ends in "nt", the per-phase timing.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PIXELNERF_ALLOW_VARIANT=1  # the variants report a negative ABI revision (tools/build_variant.sh)
shopt -s nullglob
for lib in default build/libpnr_s_*.so; do
    name=$(basename "$lib" .so); name=${name#libpnr_}
    if [ "$lib" = default ]; then unset PIXELNERF_HIP_LIB; else export PIXELNERF_HIP_LIB="$PWD/$lib"; fi
    {
        echo "=== $name"
        timeout 300 python tools/gpu_split_quickbench.py 2>&1 | grep -v amdgpu.ids
        case "$name" in *nt) ;; *) timeout 200 python tools/gpu_phase_timing_split.py 2>&1 | grep -v amdgpu.ids ;; esac
    } > "gpurun_out/sab_$name.txt" 2>&1
    head -5 "gpurun_out/sab_$name.txt"
done
