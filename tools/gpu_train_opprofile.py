#!/usr/bin/env python3
"""Which torch ops launch the small copy / fill kernels of one fp32-class training step?  torch.profiler over 4 steps,
aten::copy_ / fill_ / zero_ / to / clone grouped by Python call site."""
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
sites = collections.Counter()
real = {}
for name in ("copy_", "fill_", "zero_", "zeros", "clone", "to", "empty", "cat", "contiguous"):
    pass
import torch.utils._python_dispatch as pd
class Spy(pd.TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        n = func.__name__
        if not any(k in n for k in ("view", "reshape", "detach", "alias", "expand", "permute", "transpose", "slice", "select", "unsqueeze", "squeeze", "t.default", "stride", "size", "is_", "empty", "_unsafe_view", "as_strided", "split", "unbind", "lift")):
            st = traceback.extract_stack(limit=24)
            site = next((f"{os.path.basename(f.filename)}:{f.lineno}" for f in reversed(st)
                         if f.name not in ("__torch_dispatch__", "_wrapped") and ("pixel-nerf_amd" in f.filename or "pixelnerf_amd" in f.filename or "gpu_train_opprofile" in f.filename)), "torch-internal")
            sites[(n, site)] += 1
        return func(*args, **(kwargs or {}))
N = 4
# the dispatch mode is thread-local and the backward of the render Function runs on autograd's device thread: enter it there too
from pixelnerf_amd import autograd as _ag
for _fn in (_ag._RenderFunction,):
    _orig = _fn.backward
    def _wrapped(ctx, *g, _orig=_orig):
        with Spy():
            return _orig(ctx, *g)
    _fn.backward = staticmethod(_wrapped)
with Spy():
    for _ in range(N): step()
torch.cuda.synchronize()
print(f"aten ops per step (precision {prec}), by call site:")
for (n, site), c in sorted(sites.items(), key=lambda kv: -kv[1]):
    print(f"  {c / N:6.1f}  {n:28s} {site}")
