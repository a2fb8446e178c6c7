#!/usr/bin/env python3
"""BASELINE config (5): sn64 training step, 4 objects x 128 rays, 64 coarse + 32 fine (16 depth),
forward + backward (+ Adam) on one MI355X.  HIP path vs eager PyTorch-ROCm autograd through the
oracle restatement of the reference (fp32), same inputs."""
import os
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
    rays = synthetic.target_rays(meta, n_rays=128).to(dev)  # (4,128,8)
    gt = torch.rand(4, 128, 3, device=dev)
    mc, mf = synthetic.make_mlp_params(11), synthetic.make_mlp_params(12)
    quick = "--quick" in sys.argv  # profiling runs: one precision, no eager baseline
    for prec in (("f16",) if quick else ("f16", "bf16", "f16")):
        net = make_model(default_model_conf(), precision=prec).to(dev).train()
        net.mlp_coarse.load_state_dict(mc)
        net.mlp_fine.load_state_dict(mf)
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
        # as bench.py: PyTorch's single-kernel form of the same optimizer (PNR_ADAM_FOREACH=1: the default foreach form)
        opt = torch.optim.Adam(params, lr=1e-4, fused=not os.environ.get("PNR_ADAM_FOREACH"))

        def step():
            rd = DotMap(render_par(rays, want_weights=True))
            loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            lat.grad = None
            loss.backward()
            opt.step()
            return loss

        first = step().item()
        for _ in range(7):
            step()
        torch.cuda.synchronize()
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            loss = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"HIP {prec}: {dt*1e3:8.2f} ms/step  {1/dt:7.2f} steps/s  {512/dt:9.0f} rays/s   loss {first:.5f} -> {loss.item():.5f}", flush=True)

    if quick:
        return
    # eager PyTorch-ROCm autograd baseline (oracle restatement, fp32)
    from oracle import pnr_oracle as O
    O.USE_GRID_SAMPLE = True
    sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    sc["latent"] = sc["latent"].clone().requires_grad_(True)
    pc = {k: v.to(dev).clone().requires_grad_(True) for k, v in mc.items()}
    pf = {k: v.to(dev).clone().requires_grad_(True) for k, v in mf.items()}
    opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=1e-4)

    def estep():
        noise = {"u1": torch.rand(512, 64, device=dev), "u2": torch.rand(512, 16, device=dev),
                 "u3": torch.rand(512, 16, device=dev), "n4": torch.randn(512, 16, device=dev)}
        out = O.render(sc, pc, pf, rays, noise, 64, 32, 16, white_bkgd=True)
        loss = ((out["coarse"]["rgb"] - gt) ** 2).mean() + ((out["fine"]["rgb"] - gt) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        sc["latent"].grad = None
        loss.backward()
        opt.step()
        return loss

    for _ in range(3):
        estep()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        estep()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print(f"eager torch-ROCm fp32: {dt*1e3:8.2f} ms/step  {1/dt:7.2f} steps/s  {512/dt:9.0f} rays/s", flush=True)


if __name__ == "__main__":
    main()
