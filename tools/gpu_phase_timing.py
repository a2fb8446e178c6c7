#!/usr/bin/env python3
"""Per-phase time breakdown of the fused network kernel (wave 0 of workgroup 0)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops
from testdata import synthetic

dev = torch.device("cuda:0")
scene, meta = synthetic.make_scene("sn64")
sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev), scene["image_shape"], 1)
R, K = 16384, 192
rays = synthetic.target_rays(meta).reshape(-1, 8).repeat(4, 1)[:R].contiguous().to(dev)
z = torch.sort(ops.sample_coarse(rays, torch.rand(R, K, device=dev)), dim=-1)[0]
state = {k: v.to(dev) for k, v in synthetic.make_mlp_params(11).items()}
fold = "--no-fold" not in sys.argv
pk = ops.pack_mlp(state, "f16", folded=fold)
tab = ops.fold_latent(sc, state, "f16") if fold else None
print("folded stream" if fold else "full stream (--no-fold)")
for it in range(2):
    t = ops.debug_phase_timing(sc, pk, rays, z, tables=tab)
MT = int(os.environ.get('PNR_TILE', '64'))
ntile = ((R * K + MT - 1) // MT + 255) // 256
tot = [sum(v[w] for v in t.values()) for w in range(8)]
print(f"tile {MT} pts; tiles by WG0: {ntile}; per-tile ticks per wave: " + " ".join(f"{x/ntile:8.0f}" for x in tot))
print("phase          " + " ".join(f"   wave{w}" for w in range(8)) + "   (ticks per tile)")
for k, v in t.items():
    print(f"  {k:12s} " + " ".join(f"{x/ntile:8.0f}" for x in v))
