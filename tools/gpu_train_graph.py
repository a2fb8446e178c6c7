#!/usr/bin/env python3
"""HIP-graph capture of the config-5 training step (diagnostic / measurement): captures forward, forward+backward and
forward+backward+Adam in turn, replays each, and reports ms/step next to the eager-launch step.  faulthandler prints
the Python stack if a capture crashes.  Prints one JSON line at the end (bench.py runs this as a child process so that a
crash in graph capture cannot take the benchmark line down)."""
import faulthandler
import json
import os
import sys
import time

import torch

faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd.model import make_model  # noqa: E402
from pixelnerf_amd.render import NeRFRenderer  # noqa: E402
from pixelnerf_amd.util import DotMap  # noqa: E402
from pixelnerf_amd.util.conf import default_model_conf  # noqa: E402
from testdata import synthetic  # noqa: E402


def main():
    argv = list(sys.argv[1:])
    prec = "f16"
    if "--prec" in argv:  # "f16" (default) or "f16x3": the fused fp32-class step
        i = argv.index("--prec")
        prec = argv[i + 1]
        del argv[i:i + 2]
    stages = [a for a in argv if not a.startswith("-")] or ["fwd", "fwdbwd", "step"]
    dev = torch.device("cuda:0")
    scene, meta = synthetic.make_scene("train")
    rays = synthetic.target_rays(meta, n_rays=128).to(dev)
    gt = torch.rand(4, 128, 3, device=dev)
    net = make_model(default_model_conf(), precision=prec).to(dev).train()
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
    opt = torch.optim.Adam(params, lr=1e-4, capturable=True, fused=True)  # the single-kernel form, as in bench.py
    static_loss = torch.zeros((), device=dev)

    def body(stage):
        if stage == "fwd":
            with torch.no_grad():
                rd = DotMap(render_par(rays, want_weights=True))
            static_loss.copy_(((rd.fine.rgb - gt) ** 2).mean())
            return
        rd = DotMap(render_par(rays, want_weights=True))
        loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
        for p in params:
            p.grad = None
        lat.grad = None
        loss.backward()
        if stage == "step":
            opt.step()
        static_loss.copy_(loss.detach())

    res = {}
    for stage in stages:
        print("stage", stage, file=sys.stderr, flush=True)
        for _ in range(5):
            body(stage)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 20
        for _ in range(n):
            body(stage)
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / n
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                body(stage)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        print("capturing", stage, file=sys.stderr, flush=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body(stage)
        print("captured", stage, file=sys.stderr, flush=True)
        for _ in range(3):
            graph.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay()
        torch.cuda.synchronize()
        g = (time.perf_counter() - t0) / n
        res[stage] = {"eager_ms": eager * 1e3, "graph_ms": g * 1e3, "loss": float(static_loss.item())}
        print(stage, res[stage], file=sys.stderr, flush=True)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
