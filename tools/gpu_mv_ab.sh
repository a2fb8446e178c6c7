cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in default build/libpnr_mvregs.so build/libpnr_mvpark.so; do
  [ "$lib" != default ] && [ ! -f "$lib" ] && continue
  if [ "$lib" = default ]; then unset PIXELNERF_HIP_LIB; else export PIXELNERF_HIP_LIB="$PWD/$lib"; fi
  echo "=== $lib"; timeout 400 python tools/gpu_config_sweep.py dtu srn_car --f16 2>&1 | grep -v amdgpu.ids
done
