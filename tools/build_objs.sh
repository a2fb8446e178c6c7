#!/bin/bash
# Parallel per-TU build for A/B work:  tools/build_objs.sh NAME [TU=-DFLAG ...]
#   compiles every pixel-nerf_amd/csrc/pnr_*.hip to build/obj_NAME/*.o (one hipcc per TU, in parallel; a TU listed as
#   `pnr_bwd=-DPNR_X_FOO` gets that flag) with -DPNR_VARIANT and links build/libpnr_NAME.so.  Objects of TUs without a flag are
#   shared through build/obj_base/ (rebuilt when the source is newer).  Select with PIXELNERF_HIP_LIB=build/libpnr_NAME.so
#   PIXELNERF_ALLOW_VARIANT=1.  (The product library is built by __graft_entry__.build() in one hipcc call.)
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"; NAME=$1; shift
declare -A FLAGS; for kv in "$@"; do FLAGS[${kv%%=*}]="${kv#*=}"; done
mkdir -p "$REPO/build/obj_base" "$REPO/build/obj_$NAME"; cd "$REPO/pixel-nerf_amd/csrc"
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -Wno-unused-value -DPNR_VARIANT"
pids=()
for src in pnr_*.hip; do tu=${src%.hip}
  if [ -n "${FLAGS[$tu]}" ]; then $CC ${FLAGS[$tu]} $src -o "$REPO/build/obj_$NAME/$tu.o" & pids+=($!)
  else o="$REPO/build/obj_base/$tu.o"
    if [ ! -f "$o" ] || [ -n "$(find . -maxdepth 1 \( -name "$src" -o -name '*.h' \) -newer "$o")" ] || [ ../../include/pixelnerf_hip.h -nt "$o" ]; then $CC $src -o "$o" & pids+=($!); fi
    ln -sf "$o" "$REPO/build/obj_$NAME/$tu.o"; fi
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$REPO"/build/obj_$NAME/*.o -o "$REPO/build/libpnr_$NAME.so"
echo "built build/libpnr_$NAME.so ($*)"
