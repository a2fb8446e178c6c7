#!/usr/bin/env python3
"""debug: where does the fine-loss -> coarse-depth gradient of the generic-model path vanish?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hip_generic_training import TinyField, make_rays
from pixelnerf_amd import ops, autograd
dev = torch.device("cuda:0")
SB, B, Kc, Kf, Kfd, std = 2, 48, 16, 16, 8, 0.05
R = SB * B
rays3 = make_rays(SB, B, 5).to(dev); rays = rays3.reshape(-1, 8)
g = torch.Generator().manual_seed(9)
u1 = torch.rand(R, Kc, generator=g).to(dev); u2 = torch.rand(R, Kf - Kfd, generator=g).to(dev); u3 = torch.rand(R, Kf - Kfd, generator=g).to(dev)
n4 = torch.randn(R, Kfd, generator=g).to(dev)
model = TinyField(3).to(dev)
def run(z, coarse):
    K = z.shape[1]
    pts = (rays[:, None, :3] + z.unsqueeze(2) * rays[:, None, 3:6]).reshape(SB, -1, 3)
    vd = rays[:, None, 3:6].expand(-1, K, -1).reshape(SB, -1, 3)
    out = model(pts, coarse=coarse, viewdirs=vd).reshape(R, K, 4)
    return autograd.composite_autograd(rays, z, out, True)
zc = ops.sample_coarse(rays, u1)
wc, rgbc, depthc = run(zc, True)
print("depth_c range", float(depthc.min()), float(depthc.max()), "requires_grad", depthc.requires_grad)
depthc.register_hook(lambda gr: print("grad depth_c: absmax", float(gr.abs().max())))
z_all = autograd.sample_fine_autograd(rays, wc.detach(), depthc, zc, u2, u3, n4, std, False)
print("z_all requires_grad", z_all.requires_grad)
z_all.register_hook(lambda gr: print("grad z_all: absmax", float(gr.abs().max()), "nonzero", int((gr != 0).sum())))
zraw = depthc.detach().unsqueeze(1) + n4 * std
print("live fraction", float(((zraw < rays[:, 7:8]) & (zraw > rays[:, 6:7])).float().mean()))
wf, rgbf, depthf = run(z_all, False)
(rgbf ** 2).mean().backward()
gc = torch.cat([p.grad.reshape(-1) for p in model.coarse_net.parameters()])
print("coarse net grad absmax", float(gc.abs().max()))
# guard: plain vs guarded bits on mv_mini
from helpers import load_golden, mlp_params, scene_for
s, _ = scene_for("mv_mini")
sc = ops.make_scene(s["latent"].to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])
gg = load_golden("stages")
xyz = torch.from_numpy(gg["mv_mini_xyz"]).to(dev); vd = torch.from_numpy(gg["mv_mini_viewdirs"]).to(dev)
state = {k: v.to(dev) for k, v in mlp_params(11).items()}
pk, tab = ops.pack_mlp(state, "f16x3"), ops.fold_latent(sc, state, "f16x3")
a = ops.eval_points(sc, pk, xyz, vd, tables=tab).clone()
a2 = ops.eval_points(sc, pk, xyz, vd, tables=tab).clone()
ops.saturation_guard_arm(dev); b = ops.eval_points(sc, pk, xyz, vd, tables=tab).clone(); ops.saturation_guard_disarm(dev)
print("mv_mini plain vs plain equal", torch.equal(a, a2), " plain vs guarded: max abs diff", float((a - b).abs().max()), "n diff", int((a != b).sum()), "of", a.numel(),
      "bits", ops.saturation_guard_poll(dev, wait=True))
