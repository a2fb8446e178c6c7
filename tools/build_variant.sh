#!/bin/bash
# Build an A/B variant of libpixelnerf_hip.so with extra -D flags:  tools/build_variant.sh NAME [-DFLAG ...]
# -> build/libpnr_NAME.so (travels to the GPU box; select it with PIXELNERF_HIP_LIB=build/libpnr_NAME.so PIXELNERF_ALLOW_VARIANT=1).
# Every variant is compiled with -DPNR_VARIANT: the kernel sources honour an experiment switch only under
# `#if defined(PNR_VARIANT) && defined(SWITCH)`, and such a library reports a negative ABI revision that the product binding refuses.
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
mkdir -p "$REPO/build"
cd "$REPO/pixel-nerf_amd/csrc"
SRC=$(ls pnr_*.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -DPNR_VARIANT "$@" $SRC -o "$REPO/build/libpnr_$NAME.so"
echo "built build/libpnr_$NAME.so ($*)"
