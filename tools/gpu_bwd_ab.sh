#!/bin/bash
# A/B of library variants on the training-backward kernels (tools/gpu_bwd_kernels_bench.py) -> gpurun_out/bwd_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PIXELNERF_ALLOW_VARIANT=1  # the variants report a negative ABI revision (tools/build_variant.sh)
: > gpurun_out/bwd_ab.txt
shopt -s nullglob
for lib in default build/libpnr_*.so; do
    if [ "$lib" = default ]; then unset PIXELNERF_HIP_LIB; else export PIXELNERF_HIP_LIB="$PWD/$lib"; fi
    echo "=== $lib" >> gpurun_out/bwd_ab.txt
    timeout 200 python tools/gpu_bwd_kernels_bench.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/bwd_ab.txt
done
cat gpurun_out/bwd_ab.txt
