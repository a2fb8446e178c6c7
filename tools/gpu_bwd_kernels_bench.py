#!/usr/bin/env python3
"""Stand-alone timing of the training-backward kernels at BASELINE config-5 sizes (GPU box):
latent scatter, batched weight gradient, lin_out gradient."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops  # noqa: E402
from testdata import synthetic  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def main():
    dev = torch.device("cuda:0")
    scene, meta = synthetic.make_scene("train")
    sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev),
                        scene["image_shape"], scene["NS"])
    rays = synthetic.target_rays(meta, n_rays=128).reshape(-1, 8).to(dev)
    for K in (64, 96):
        z = torch.sort(ops.sample_coarse(rays, torch.rand(512, K, device=dev)), dim=-1)[0]
        P = 512 * K
        d_zlat = torch.randn(P, 512, device=dev)
        out = torch.zeros(4, 32, 32, 512, device=dev)
        us = timeit(lambda: ops.latent_scatter(sc, rays, z, d_zlat, out))
        print(f"latent_scatter  P={P:6d}: {us:8.1f} us   ({P * 2048 / us / 1e6:.2f} TB/s of d_zlat)", flush=True)
        for dt, prec in ((torch.float16, 0),):
            jobs = [(torch.randn(P, 512, device=dev).to(dt), torch.randn(P, 512, device=dev).to(dt), True, True) for _ in range(13)]
            jobs.append((jobs[0][0], torch.randn(P, 64, device=dev).to(dt), True, False, 64, 42))
            us = timeit(lambda: ops.weight_grad_batched(jobs, prec, 1.0))
            fl = 13 * 2 * P * 512 * 512
            print(f"weight_grad x14 P={P:6d}: {us:8.1f} us   {fl / us / 1e6:.0f} TFLOP/s   unique bytes {13 * P * 2048 / us / 1e6:.2f} TB/s", flush=True)
            g = torch.randn(P, 4, device=dev)
            us = timeit(lambda: ops.lin_out_grad(g, jobs[0][1], prec))
            print(f"lin_out_grad    P={P:6d}: {us:8.1f} us", flush=True)


if __name__ == "__main__":
    main()
