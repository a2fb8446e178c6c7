#!/bin/bash
# A/B of library variants on the GPU box: for every build/libpnr_*.so (plus the default library) run the
# kernel-level quick bench (f16 folded, sn64) and the per-phase timing; results -> gpurun_out/ab_<name>.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PIXELNERF_ALLOW_VARIANT=1  # the variants report a negative ABI revision (tools/build_variant.sh)
shopt -s nullglob
for lib in default build/libpnr_*.so; do
    name=$(basename "$lib" .so); name=${name#libpnr_}
    case "$name" in t64*) export PNR_TILE=64 ;; *) export PNR_TILE=96 ;; esac
    if [ "$lib" = default ]; then unset PIXELNERF_HIP_LIB; else export PIXELNERF_HIP_LIB="$PWD/$lib"; fi
    {
        echo "=== $name"
        timeout 300 python tools/gpu_quickbench.py --fold-only --sn64 2>&1 | grep -v amdgpu.ids
        case "$name" in *nt) ;; *) timeout 120 python tools/gpu_phase_timing.py 2>&1 | grep -v amdgpu.ids ;; esac
    } > "gpurun_out/ab_$name.txt" 2>&1
    tail -n +1 "gpurun_out/ab_$name.txt" | head -4
done
