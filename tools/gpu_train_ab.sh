#!/bin/bash
# A/B of library variants on the config-5 training step (tools/gpu_train_bench.py --quick) -> gpurun_out/train_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export PIXELNERF_ALLOW_VARIANT=1  # the variants report a negative ABI revision (tools/build_variant.sh)
: > gpurun_out/train_ab.txt
shopt -s nullglob
for lib in default build/libpnr_*.so default; do
    if [ "$lib" = default ]; then unset PIXELNERF_HIP_LIB; else export PIXELNERF_HIP_LIB="$PWD/$lib"; fi
    echo "=== $lib" >> gpurun_out/train_ab.txt
    timeout 200 python tools/gpu_train_bench.py --quick 2>&1 | grep -v amdgpu.ids >> gpurun_out/train_ab.txt
done
cat gpurun_out/train_ab.txt
