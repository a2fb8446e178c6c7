import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pixelnerf_amd import ops
from testdata import synthetic
dev = torch.device("cuda:0")
s, meta = synthetic.make_scene("train")
sc = ops.make_scene(s["latent"].to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])
rays = synthetic.target_rays(meta, n_rays=128).reshape(-1, 8).to(dev)
K = 96
z = ops.sample_coarse(rays, torch.rand(rays.shape[0], K, device=dev))
d = torch.randn(s["NS"] * rays.shape[0] * K, 512, device=dev)
out = torch.zeros(4, 32, 32, 512, device=dev)
for i in range(3):
    print("call", i, flush=True)
    ops.latent_scatter(sc, rays, z, d, out); torch.cuda.synchronize()
