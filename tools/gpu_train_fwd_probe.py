#!/usr/bin/env python3
"""Probe for the training forward at BASELINE config-5 sizes: per-point network launch unfolded / folded / with training
dumps, and the cost of folding lin_z into the 4 x 32 x 32 grid (what a train-through-the-tables step would pay per step)."""
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
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    scene, meta = synthetic.make_scene("train")
    sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev),
                        scene["image_shape"], scene["NS"])
    state = {k: v.to(dev) for k, v in synthetic.make_mlp_params(11).items()}
    rays = synthetic.target_rays(meta, n_rays=128).reshape(-1, 8).to(dev)
    pk, pkf = ops.pack_mlp(state, "f16"), ops.pack_mlp(state, "f16", folded=True)
    print(f"fold_latent (4 x 32 x 32 grid, 3 tables): {timeit(lambda: ops.fold_latent(sc, state, 'f16')):8.1f} us")
    print(f"pack_mlp: {timeit(lambda: ops.pack_mlp(state, 'f16')):8.1f} us   pack_mlp bwd: {timeit(lambda: ops.pack_mlp(state, 'f16', backward=True)):8.1f} us")
    tab = ops.fold_latent(sc, state, "f16")
    for K in (64, 96):
        z = torch.sort(ops.sample_coarse(rays, torch.rand(512, K, device=dev)), dim=-1)[0]
        a = timeit(lambda: ops.eval_ray_samples(sc, pk, rays, z))
        b = timeit(lambda: ops.eval_ray_samples(sc, pkf, rays, z, tables=tab))
        c = timeit(lambda: ops.eval_ray_samples_train(sc, pk, rays, z))
        print(f"P={512 * K}: unfolded {a:7.1f} us   folded {b:7.1f} us   train (dumps) {c:7.1f} us")


if __name__ == "__main__":
    main()
