#!/usr/bin/env python3
"""Quick kernel-level timing of the fused network (GPU box): points/s and TFLOP/s."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops  # noqa: E402
from testdata import synthetic  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cases = (("sn64", 16384, 64), ("sn64", 16384, 192), ("srn_car", 8192, 192))
    if "--sn64" in sys.argv:
        cases = (("sn64", 16384, 64), ("sn64", 16384, 192))
    if "--srn" in sys.argv:
        cases = (("srn_car", 8192, 192),)
    for scene_name, R, K in cases:
        scene, meta = synthetic.make_scene(scene_name)
        NS = scene["NS"]
        sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev),
                            scene["c"].to(dev), scene["image_shape"], NS)
        rays = synthetic.target_rays(meta).reshape(-1, 8)
        rays = rays.repeat((R + rays.shape[0] - 1) // rays.shape[0], 1)[:R].contiguous().to(dev)
        u = torch.rand(R, K, device=dev)
        z = ops.sample_coarse(rays, u)
        z, _ = torch.sort(z, dim=-1)
        combos = (("f16", False), ("f16", True), ("bf16", False), ("bf16", True))
        if "--fold-only" in sys.argv:
            combos = (("f16", True), ("bf16", True)) if "--bf16" in sys.argv else (("f16", True),)
        for prec, fold in combos:
            state = {k: v.to(dev) for k, v in synthetic.make_mlp_params(11).items()}
            pk = ops.pack_mlp(state, prec, folded=fold)
            tab = ops.fold_latent(sc, state, prec) if fold else None
            for _ in range(2):
                ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 3
            for _ in range(n):
                ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            flop_pt = (4.7616e6 * NS + 2.1012e6)  # algorithmic (the folded form executes 3 x 0.524 MFLOP fewer per view)
            print(f"{scene_name} NS={NS} R={R} K={K} {prec}{' folded' if fold else ''}: {dt*1e3:8.2f} ms  {R*K/dt/1e6:8.2f} Mpts/s  "
                  f"{R*K*flop_pt/dt/1e12:8.1f} TFLOP/s (algorithmic)", flush=True)


if __name__ == "__main__":
    main()
