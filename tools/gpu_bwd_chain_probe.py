#!/usr/bin/env python3
"""Stand-alone timing of the fused data-gradient chain (pnr_mlp_backward) and the training forward at BASELINE config-5
sizes; used with library variants (tools/build_variant.sh + PIXELNERF_HIP_LIB) to see what each part of the chain costs."""
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
    pk, pkb = ops.pack_mlp(state, "f16"), ops.pack_mlp(state, "f16", backward=True)
    for K in (64, 96):
        z = torch.sort(ops.sample_coarse(rays, torch.rand(512, K, device=dev)), dim=-1)[0]
        rgbs, dumps = ops.eval_ray_samples_train(sc, pk, rays, z)
        g = torch.randn(512 * K, 4, device=dev)
        fwd = timeit(lambda: ops.eval_ray_samples_train(sc, pk, rays, z))
        bwd = timeit(lambda: ops.mlp_backward(pkb, dumps, g, 1.0))
        print(f"P={512 * K}: train forward {fwd:7.1f} us   data-gradient chain {bwd:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
