#!/usr/bin/env python3
"""Encoder output formatting on the GPU box: one HIP pass vs the torch
interpolate + cat + NCHW->NHWC transpose it replaces, against the 8 TB/s HBM roofline."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops  # noqa: E402
from testdata import synthetic  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    dev = torch.device("cuda:0")
    stages = [t.to(dev) for t in synthetic.pyramid_stages("dtu")]
    src = sum(t.numel() for t in stages) * 4
    NV, H0, W0 = stages[0].shape[0], stages[0].shape[2], stages[0].shape[3]
    out = NV * H0 * W0 * 512 * 4

    def torch_path():
        lat = torch.cat([torch.nn.functional.interpolate(t, (H0, W0), mode="bilinear", align_corners=True) for t in stages], 1)
        return ops.nchw_to_nhwc(lat)

    for name, fn, by in (("HIP one pass, NHWC + NCHW", lambda: ops.pyramid_to_latent(stages, True), src + 2 * out),
                         ("HIP one pass, NHWC only", lambda: ops.pyramid_to_latent(stages, False), src + out),
                         ("torch interpolate + cat + HIP transpose", torch_path, src + out)):
        dt = timeit(fn)
        print(f"{name:42s}: {dt * 1e3:7.3f} ms   algorithmic {by / 1e6:6.1f} MB -> {by / dt / 1e9:7.1f} GB/s = {by / dt / 8e12:.2f} of 8 TB/s",
              flush=True)


if __name__ == "__main__":
    main()
