#!/usr/bin/env python3
"""pnr_latent_scatter (d z_lat -> d feature grid) on the training workloads: config 5 (4 objects x 128 rays, 32x32 grids) coarse (64 samples)
and fine (96) passes, the 2 x 2-view scene, an srn-sized 64x64 grid; HIP events, us per call (segment pre-pass included).
A/B against another library: PIXELNERF_HIP_LIB=build/libpnr_<name>.so PIXELNERF_ALLOW_VARIANT=1 (profiles/r06_scatter_notes.md)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops
from testdata import synthetic
dev = torch.device("cuda:0")
tag = os.path.basename(os.environ.get("PIXELNERF_HIP_LIB", "product"))
gen = torch.Generator().manual_seed(3)
torch.manual_seed(3)
for name, K, hw in (("train", 64, None), ("train", 96, None), ("train_mv", 96, None), ("train", 96, 64), ("dtu", 96, None), ("dtu_train4", 96, None)):
    big = name == "dtu_train4"  # the reference's DTU training batch: twelve 150 x 200 grids, drawn on the device
    s, meta = synthetic.make_scene(name, with_latent=not big)
    if big:
        lat = torch.randn(s["SB"] * s["NS"], 512, meta["Hl"], meta["Wl"], device=dev)
    else:
        lat = s["latent"] if hw is None else torch.randn(s["latent"].shape[0], 512, hw, hw, generator=gen)
    sc = ops.make_scene(lat.to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])
    rays = synthetic.target_rays(meta, n_rays=128).reshape(-1, 8).to(dev)
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], K, device=dev))
    d = torch.randn(s["NS"] * rays.shape[0] * K, 512, device=dev)
    out = torch.zeros(lat.shape[0], lat.shape[2], lat.shape[3], 512, device=dev)
    for _ in range(3):
        ops.latent_scatter(sc, rays, z, d, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record()
    for _ in range(n):
        ops.latent_scatter(sc, rays, z, d, out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    a = torch.zeros_like(out); ops.latent_scatter(sc, rays, z, d, a)
    b = torch.zeros_like(out); ops.latent_scatter(sc, rays, z, d, b)
    print(f"scatter[{tag}] {name:8s} K={K:3d} grid {lat.shape[0]}x{lat.shape[2]}x{lat.shape[3]}: {us:8.1f} us per call "
          f"({d.numel() * 4 / us / 1e6:6.2f} TB/s of gradient rows)  bit-reproducible={bool(torch.equal(a, b))}  checksum {a.double().sum().item():.9e}", flush=True)
