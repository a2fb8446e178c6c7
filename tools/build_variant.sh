#!/bin/bash
# Build an A/B variant of libpixelnerf_hip.so with extra -D flags:  tools/build_variant.sh NAME [-DFLAG ...]
# -> build/libpnr_NAME.so (travels to the GPU box; select it with PIXELNERF_HIP_LIB=build/libpnr_NAME.so).
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
NAME=$1; shift
mkdir -p "$REPO/build"
cd "$REPO/pixel-nerf_amd/csrc"
SRC=$(ls pnr_*.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value "$@" $SRC -o "$REPO/build/libpnr_$NAME.so"
echo "built build/libpnr_$NAME.so ($*)"
