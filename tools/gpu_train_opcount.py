#!/usr/bin/env python3
"""Every aten op, memcpy and kernel of one fp32-class training step by count (torch.profiler, all threads)."""
import os, sys, collections, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd.model import make_model
from pixelnerf_amd.render import NeRFRenderer
from pixelnerf_amd.util import DotMap
from pixelnerf_amd.util.conf import default_model_conf
from testdata import synthetic
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
dev = torch.device("cuda:0")
scene, meta = synthetic.make_scene("train")
rays = synthetic.target_rays(meta, n_rays=128).to(dev)
gt = torch.rand(4, 128, 3, device=dev)
net = make_model(default_model_conf(), precision=prec).to(dev).train()
net.mlp_coarse.load_state_dict(synthetic.make_mlp_params(11)); net.mlp_fine.load_state_dict(synthetic.make_mlp_params(12))
lat = scene["latent"].to(dev).clone().requires_grad_(True)
net.encoder.latent = lat
ls = torch.tensor([32.0, 32.0], device=dev); net.encoder.latent_scaling = ls / (ls - 1) * 2.0
net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).to(dev)
render_par = rend.bind_parallel(net, None, simple_output=False).train()
opt = torch.optim.Adam(list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters()), lr=1e-4, fused=True)
def step():
    rd = DotMap(render_par(rays, want_weights=True))
    loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
    opt.zero_grad(set_to_none=True); lat.grad = None
    loss.backward(); opt.step()
for _ in range(6): step()
torch.cuda.synchronize()
# every aten op / memcpy / kernel of N steps, all threads (the autograd engine's device thread included), by count
from torch.profiler import profile, ProfilerActivity
N = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N): step()
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.count)
print(f"events per step (precision {prec}), count >= 1 per step:")
for e in rows:
    if e.count >= N:
        dev_us = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
        print(f"  {e.count / N:7.1f}  cpu {e.cpu_time_total / N:8.1f} us  dev {dev_us / N:8.1f} us  {e.key[:110]}")
