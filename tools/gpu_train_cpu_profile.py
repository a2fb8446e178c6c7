#!/usr/bin/env python3
"""Host-side cost of the config-5 training step: cProfile over 30 steps (top cumulative entries) and the enqueue-only time
per step (no synchronisation between steps: what the host needs to issue a step when the GPU never makes it wait)."""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from testdata import synthetic  # noqa: E402
from pixelnerf_amd.model import make_model  # noqa: E402
from pixelnerf_amd.render import NeRFRenderer  # noqa: E402
from pixelnerf_amd.util import DotMap  # noqa: E402
from pixelnerf_amd.util.conf import default_model_conf  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    scene, meta = synthetic.make_scene("train")
    rays = synthetic.target_rays(meta, n_rays=128).to(dev)
    gt = torch.rand(4, 128, 3, device=dev)
    net = make_model(default_model_conf(), precision="f16").to(dev).train()
    net.mlp_coarse.load_state_dict(synthetic.make_mlp_params(11))
    net.mlp_fine.load_state_dict(synthetic.make_mlp_params(12))
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([32.0, 32.0], device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).to(dev)
    render_par = rend.bind_parallel(net, None, simple_output=False).train()
    params = list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters())
    opt = torch.optim.Adam(params, lr=1e-4, fused=True)

    def step():
        rd = DotMap(render_par(rays, want_weights=True))
        loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        lat.grad = None
        loss.backward()
        opt.step()

    for _ in range(8):
        step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_enq = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"enqueue {t_enq * 1e3:.2f} ms/step (host only), {t_all * 1e3:.2f} ms/step with the GPU drained")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(38)
    print("\n".join(line[:150] for line in s.getvalue().splitlines()))


if __name__ == "__main__":
    main()
