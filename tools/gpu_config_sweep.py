#!/usr/bin/env python3
"""Full-size single-GPU runs of BASELINE configs (2) sn64, (3) srn_car, (4) DTU: rays/s through
render_par(rays) and PSNR of a 256-ray sample against the CPU oracle (identical noise)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import pnr_oracle as O  # noqa: E402
from testdata import synthetic  # noqa: E402

FLOP_V, FLOP_P = 4.7616e6, 2.1012e6


def main():
    dev = torch.device("cuda:0")
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    precs = ("f16",) if "--f16" in sys.argv else ("f16x3",) if "--f16x3" in sys.argv else ("f16", "bf16")
    for scene_name, n_img in (("sn64", 16), ("srn_car", 4), ("dtu", 1)):
        if only and scene_name not in only:
            continue
        for prec in precs:
            scene, meta, net, renderer, mlps = bench.build(dev, prec, scene_name)
            rays = synthetic.target_rays(meta).reshape(-1, 8)
            rays = rays.repeat(n_img, 1).contiguous().to(dev)
            R = rays.shape[0]
            render_par = renderer.bind_parallel(net, None, simple_output=True).eval()
            with torch.no_grad():
                for _ in range(2):
                    render_par(rays[None])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 3
                for _ in range(n):
                    render_par(rays[None])
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
                NS = scene["NS"]
                fl = 256 * (FLOP_V * NS + FLOP_P)
                # parity spot check on 256 rays with explicit noise
                idx = torch.randperm(R // n_img, generator=torch.Generator().manual_seed(1))[:256]
                rs = rays[: R // n_img][idx.to(dev)].cpu()
                noise = synthetic.make_noise(256, 64, 128, 16, seed=5)
                out = renderer(net, rs.to(dev)[None], _noise={k: v.to(dev) for k, v in noise.items()})
                ref = O.render(scene, mlps[0], mlps[1], rs[None], noise, 64, 128, 16, white_bkgd=meta["white_bkgd"])
                ps = O.psnr(out.fine.rgb.cpu(), ref["fine"]["rgb"])
            print(f"{scene_name:8s} NS={NS} {meta['W']}x{meta['H']} grid {tuple(scene['latent'].shape)} {prec:5s}: R={R:7d} "
                  f"{dt*1e3:8.1f} ms  {R/dt/1e3:8.1f} k rays/s  {R*fl/dt/1e12:7.1f} TFLOP/s  PSNR(256 rays) {ps:5.1f} dB", flush=True)
            del net, scene
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
