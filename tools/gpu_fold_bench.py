#!/usr/bin/env python3
"""Per-scene fold of lin_z into fp32 tables (pnr_fold_latent_f32 = fold_split_kernel) at the three grid sizes of BASELINE:
sn64 (1x32x32 texels), config-5 training (4x32x32), srn_car (2x64x64), DTU (3x150x200).  HIP events, per network."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops
from testdata import synthetic
dev = torch.device("cuda:0")
state = {k: v.to(dev) for k, v in synthetic.make_mlp_params(11).items()}
for name in ("sn64", "train", "srn_car", "dtu"):
    s, meta = synthetic.make_scene(name)
    sc = ops.make_scene(s["latent"].to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])
    for _ in range(3):
        t = ops.fold_latent(sc, state, "f16x3")
    torch.cuda.synchronize()
    n = 50 if name != "dtu" else 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        t = ops.fold_latent(sc, state, "f16x3")
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    texels = s["latent"].shape[0] * s["latent"].shape[2] * s["latent"].shape[3]
    tf = texels * 3 * 2 * 512 * 512 * 3 / (us * 1e-6) / 1e12
    print(f"fold {name:8s} {texels:7d} texels: {us:9.1f} us per network  ({tf:7.1f} TFLOP/s of executed f16 MFMAs)")
