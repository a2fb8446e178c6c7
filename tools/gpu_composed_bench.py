#!/usr/bin/env python3
"""Rate of the COMPOSED path (one HIP operator per nn.Linear) next to the fused kernels, same GPU, same point count:
PixelNeRFNet.forward on (1, P, 3) points of the sn64 scene -- the shipped conf (fused fp32-class kernel) against the reference's
DEFAULT code arrangement `use_code_viewdirs=True` (d_in = 78: composed path, same 512 x 5 MLP), inference and forward + backward."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from testdata import synthetic
from pixelnerf_amd.model import make_model
from pixelnerf_amd.util.conf import Conf, default_model_conf
dev = torch.device("cuda:0")
scene, meta = synthetic.make_scene("sn64", seed=2)

def build(conf, prec="f16x3"):
    net = make_model(conf, precision=prec).to(dev).eval()
    for i, mlp in enumerate((net.mlp_coarse, net.mlp_fine)):
        shapes = [(k, tuple(v.shape)) for k, v in mlp.state_dict().items()]
        mlp.load_state_dict({k: v.to(dev) for k, v in synthetic.fill_state(shapes, 31 + i).items()})
    lat = scene["latent"].to(dev)
    net.encoder.latent = lat
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32, device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    return net

def timeit(fn, n=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

P = 1 << 18
g = torch.Generator().manual_seed(0)
xyz = (torch.rand(1, P, 3, generator=g) * 2 - 1).to(dev)
vd = torch.nn.functional.normalize(torch.randn(1, P, 3, generator=g), dim=-1).to(dev)
variant = Conf(dict(default_model_conf(), use_code_viewdirs=True))
for name, conf in (("shipped conf (fused kernels)", default_model_conf()), ("use_code_viewdirs=True (composed path)", variant)):
    net = build(conf)
    assert net.fused_supported() == name.startswith("shipped")
    with torch.no_grad():
        t = timeit(lambda: net(xyz, coarse=True, viewdirs=vd))
    print(f"{name}: inference {P / t / 1e6:7.2f} M points/s ({t * 1e3:.2f} ms for {P} points)", flush=True)
    net.train()
    Pb = 1 << 16
    def step():
        for p in net.parameters(): p.grad = None
        out = net(xyz[:, :Pb], coarse=True, viewdirs=vd[:, :Pb])
        out.square().mean().backward()
    t = timeit(step, n=4, warm=2)
    print(f"{name}: forward + backward {Pb / t / 1e6:7.2f} M points/s ({t * 1e3:.2f} ms for {Pb} points)", flush=True)
