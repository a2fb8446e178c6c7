#!/usr/bin/env python3
"""Kernel-level timing of the fp32-class fused network (precision f16x3) on the three scene shapes: Mpts/s, TFLOP/s."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops  # noqa: E402
from testdata import synthetic  # noqa: E402

dev = torch.device("cuda:0")
cases = [("sn64", 16384, 192), ("srn_car", 8192, 192), ("dtu", 8192, 192)]
if "--mv" in sys.argv:
    cases = cases[1:]
for scene_name, R, K in cases:
    scene, meta = synthetic.make_scene(scene_name)
    NS = scene["NS"]
    sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev), scene["image_shape"], NS)
    rays = synthetic.target_rays(meta).reshape(-1, 8)
    rays = rays.repeat((R + rays.shape[0] - 1) // rays.shape[0], 1)[:R].contiguous().to(dev)
    z = torch.sort(ops.sample_coarse(rays, torch.rand(R, K, device=dev)), dim=-1)[0]
    state = {k: v.to(dev) for k, v in synthetic.make_mlp_params(11).items()}
    pk = ops.pack_mlp(state, "f16x3")
    tab = ops.fold_latent(sc, state, "f16x3")
    for _ in range(2):
        ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    flop_pt = 4.7616e6 * NS + 2.1012e6
    print(f"{scene_name} NS={NS} R={R} K={K} f16x3: {dt*1e3:8.2f} ms  {R*K/dt/1e6:8.2f} Mpts/s  {R*K*flop_pt/dt/1e12:8.1f} TFLOP/s (algorithmic)  "
          f"-> {R*K/dt/256/1e3:7.1f} k rays/s at 256 evals/ray", flush=True)
    del tab, pk, sc
    torch.cuda.empty_cache()
