#!/usr/bin/env python3
"""Gradient parity table: HIP training path vs torch autograd through the CPU oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # (this file's own directory: helpers.py)
from helpers import golden_setup, mlp_params  # noqa: E402
from oracle import pnr_oracle as O  # noqa: E402


def oracle_grads(scene, mc, mf, rays, noise, Kc, Kf, Kfd, white, lindisp, gt, detach_depth):
    sc = dict(scene)
    sc["latent"] = scene["latent"].clone().requires_grad_(True)
    pc = {k: v.clone().requires_grad_(True) for k, v in mc.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in mf.items()}
    out = O.render(sc, pc, pf, rays, noise, Kc, Kf, Kfd, white_bkgd=white, lindisp=lindisp, detach_depth=detach_depth)
    loss = ((out["coarse"]["rgb"] - gt) ** 2).mean() + ((out["fine"]["rgb"] - gt) ** 2).mean() \
        + 0.1 * out["fine"]["depth"].mean() + 0.05 * (out["coarse"]["weights"] ** 2).mean()
    loss.backward()
    return loss.item(), {"latent": sc["latent"].grad, **{"c." + k: v.grad for k, v in pc.items()},
                         **{"f." + k: v.grad for k, v in pf.items()}}


def hip_grads(dev, scene, mc, mf, rays, noise, Kc, Kf, Kfd, white, lindisp, gt, prec):
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util.conf import default_model_conf
    net = make_model(default_model_conf(), precision=prec).to(dev).train()
    net.mlp_coarse.load_state_dict(mc)
    net.mlp_fine.load_state_dict(mf)
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32, device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, white_bkgd=white, lindisp=lindisp).to(dev).train()
    out = rend(net, rays.to(dev), want_weights=True, _noise={k: v.to(dev) for k, v in noise.items()})
    g = gt.to(dev)
    loss = ((out.coarse.rgb - g) ** 2).mean() + ((out.fine.rgb - g) ** 2).mean() \
        + 0.1 * out.fine.depth.mean() + 0.05 * (out.coarse.weights ** 2).mean()
    loss.backward()
    gr = {"latent": lat.grad.cpu()}
    gr.update({"c." + k: v.grad.cpu() for k, v in net.mlp_coarse.named_parameters()})
    gr.update({"f." + k: v.grad.cpu() for k, v in net.mlp_fine.named_parameters()})
    return loss.item(), gr


def compare(name, prec="f16", verbose=True):
    dev = torch.device("cuda:0")
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    white, lindisp = bool(g["white_bkgd"]), bool(g["lindisp"])
    gt = torch.rand(rays.shape[0], rays.shape[1], 3, generator=torch.Generator().manual_seed(4))
    lo, go = oracle_grads(scene, mc, mf, rays, noise, Kc, Kf, Kfd, white, lindisp, gt, True)
    lf, gfull = oracle_grads(scene, mc, mf, rays, noise, Kc, Kf, Kfd, white, lindisp, gt, False)
    lh, gh = hip_grads(dev, scene, mc, mf, rays, noise, Kc, Kf, Kfd, white, lindisp, gt, prec)
    worst = 0.0
    rows = []
    for k in go:
        a, b, c = gh[k].double(), gfull[k].double(), go[k].double()  # hip, reference semantics, depth-detached
        rel = float((a - b).norm() / (b.norm() + 1e-30))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-30))
        rel_full = float((b - c).norm() / (b.norm() + 1e-30))
        rows.append((k, float(b.norm()), rel, cos, rel_full))
        worst = max(worst, rel)
    if verbose:
        print(f"== {name} {prec}: loss oracle {lo:.6f} hip {lh:.6f}")
        for k, n, rel, cos, rf in rows:
            print(f"  {k:28s} |g| {n:10.3e}  rel.err {rel:9.2e}  cos {cos:.6f}   (size of the depth-sample position term: {rf:8.2e})")
    return rows, (lo, lh)


if __name__ == "__main__":
    for nm in ("train_64_32", "srn_mini_64_128"):
        for prec in ("f16", "bf16"):
            compare(nm, prec)
