#!/usr/bin/env python3
"""Error levels of the exact-fp32 path against the reference goldens, and its throughput (GPU box)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import RENDER_SCENARIOS, golden_setup, load_golden, mlp_params, scene_for  # noqa: E402
from pixelnerf_amd import ops  # noqa: E402
from testdata import synthetic  # noqa: E402


def dscene(name, dev):
    s, _ = scene_for(name)
    return ops.make_scene(s["latent"].to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])


def main():
    dev = torch.device("cuda:0")
    g = load_golden("stages")
    pk = {s: ops.pack_mlp({k: v.to(dev) for k, v in mlp_params(s).items()}, "f32") for s in (11, 12)}
    for name in ("sn64", "dtu_mini", "mv_mini"):
        sc = dscene(name, dev)
        for which, seed in (("coarse", 11), ("fine", 12)):
            out = ops.eval_points(sc, pk[seed], torch.from_numpy(g[f"{name}_xyz"]).to(dev),
                                  torch.from_numpy(g[f"{name}_viewdirs"]).to(dev)).cpu().numpy()
            ref = g[f"{name}_out_{which}"]
            print(f"points {name:9s} {which:6s} rgb max {np.abs(out[..., :3] - ref[..., :3]).max():.2e}  sigma rel "
                  f"{(np.abs(out[..., 3] - ref[..., 3]) / np.maximum(1, ref[..., 3])).max():.2e}", flush=True)
    for name in RENDER_SCENARIOS:
        gg, scene, meta, mc, mf, rays, noise = golden_setup(name)
        Kc, Kf, Kfd = int(gg["n_coarse"]), int(gg["n_fine"]), int(gg["n_fine_depth"])
        sc = dscene(str(gg["scene"]), dev)
        pc = pk.get(int(gg["mlp_seed_coarse"])) or ops.pack_mlp({k: v.to(dev) for k, v in mc.items()}, "f32")
        pf = None if mf is None else (pk.get(int(gg["mlp_seed_fine"])) or ops.pack_mlp({k: v.to(dev) for k, v in mf.items()}, "f32"))
        out = ops.render_forward(sc, pc, pf, rays.reshape(-1, 8).to(dev), Kc, Kf, Kfd, {k: v.to(dev) for k, v in noise.items()},
                                 depth_std=float(gg["depth_std"]), white_bkgd=bool(gg["white_bkgd"]), lindisp=bool(gg["lindisp"]),
                                 want_weights=True)
        for p in ["coarse"] + (["fine"] if Kf else []):
            e = np.abs(out[p]["rgb"].cpu().numpy() - gg[f"{p}_rgb"].reshape(-1, 3))
            ed = np.abs(out[p]["depth"].cpu().numpy() - gg[f"{p}_depth"].reshape(-1))
            mse = float((e.astype(np.float64) ** 2).mean())
            print(f"render {name:22s} {p:6s} rgb max {e.max():.2e} frac>2e-5 {(e > 2e-5).mean():.4f}  depth max {ed.max():.2e} "
                  f"psnr {-10 * np.log10(max(mse, 1e-30)):.1f} dB", flush=True)
    # throughput
    for scene_name, R, K in (("sn64", 4096, 192), ("srn_car", 4096, 192)):
        s, meta = scene_for(scene_name)
        sc = dscene(scene_name, dev)
        rays = synthetic.target_rays(meta).reshape(-1, 8)[:R].contiguous().to(dev)
        z = torch.sort(ops.sample_coarse(rays, torch.rand(R, K, device=dev)), dim=-1)[0]
        for _ in range(2):
            ops.eval_ray_samples(sc, pk[11], rays, z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ops.eval_ray_samples(sc, pk[11], rays, z)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        flop_pt = 4.7616e6 * s["NS"] + 2.1012e6
        print(f"f32 {scene_name} R={R} K={K}: {dt * 1e3:.1f} ms  {R * K / dt / 1e6:.2f} Mpts/s  {R * K * flop_pt / dt / 1e12:.1f} TFLOP/s",
              flush=True)


if __name__ == "__main__":
    main()
