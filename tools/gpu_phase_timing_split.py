#!/usr/bin/env python3
"""Per-phase cycle breakdown of the fp32-class fused kernel (eval_split_kernel, 64-point tiles), all waves of workgroup 0."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops
from testdata import synthetic

dev = torch.device("cuda:0")
MT = 64
scene, meta = synthetic.make_scene("sn64")
sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev), scene["image_shape"], 1)
R, K = 16384, 192
rays = synthetic.target_rays(meta).reshape(-1, 8).repeat(4, 1)[:R].contiguous().to(dev)
z = torch.sort(ops.sample_coarse(rays, torch.rand(R, K, device=dev)), dim=-1)[0]
state = {k: v.to(dev) for k, v in synthetic.make_mlp_params(11).items()}
pk = ops.pack_mlp(state, "f16x3")
tab = ops.fold_latent(sc, state, "f16x3")
for _ in range(2):
    ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"f16x3 sn64 R={R} K={K}: {dt*1e3:.2f} ms  {R*K/dt/1e6:.2f} Mpts/s  {R*K*6.863e6/dt/1e12:.1f} TFLOP/s algorithmic, "
      f"{3*0.771*R*K*6.863e6/dt/1e12:.1f} executed")
if "--no-phases" not in sys.argv:
    for it in range(2):
        t = ops.debug_phase_timing_split(sc, pk, rays, z, tab)
    ntile = ((R * K + MT - 1) // MT + 255) // 256
    tot = [sum(v[w] for v in t.values()) for w in range(8)]
    print(f"tile {MT} pts; tiles by WG0: {ntile}; per-tile cycles per wave: " + " ".join(f"{x/ntile:8.0f}" for x in tot))
    print("phase          " + " ".join(f"   wave{w}" for w in range(8)) + "   (cycles per tile)")
    for k, v in t.items():
        print(f"  {k:12s} " + " ".join(f"{x/ntile:8.0f}" for x in v))
