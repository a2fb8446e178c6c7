"""diagnostic: is SpatialEncoder.forward reproducible when other kernels run between two encodes of the same image?"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pixelnerf_amd import ops
from testdata import synthetic
dev = torch.device("cuda:0")
scene, meta, net, renderer, mlps = bench.build(dev, "f16x3", "sn64")
H = W = 64
rs = np.random.RandomState(7)
img = torch.from_numpy(rs.uniform(-1, 1, (1, 3, H, W)).astype(np.float32)).to(dev)
src_pose = synthetic.pose_spherical(30.0, -20.0, meta["radius"])[None].to(dev)
focal = torch.tensor(meta["focal"][0], dtype=torch.float32)[None].to(dev)
c = torch.tensor(meta["c"], dtype=torch.float32)[None].to(dev)
def cmp(x, y): return "equal" if torch.equal(x, y) else "max|d| %.3e in %d of %d" % (float((x - y).abs().max()), int((x != y).sum()), x.numel())
def enc():
    net.encode(img.unsqueeze(0), src_pose.unsqueeze(0), focal, c=c)
    torch.cuda.synchronize()
    return [l.clone() for l in net.encoder.latents] + [net.encoder.latent.clone()]
rays = synthetic.target_rays(meta).reshape(1, -1, 8).to(dev)
with torch.no_grad():
    for mode in ("graph", "eager"):
        type(net.encoder).use_graph = mode == "graph"
        a = enc()
        b = enc()
        print(mode, "back to back :", [cmp(x, y) for x, y in zip(a, b)], flush=True)
        renderer(net, rays)          # a render in between (LDS / caches / allocator touched)
        junk = torch.randn(64, 1024, 1024, device=dev); del junk
        c2 = enc()
        print(mode, "after a render:", [cmp(x, y) for x, y in zip(a, c2)], flush=True)
        big = torch.full((256, 1024, 1024), float("nan"), device=dev); del big; torch.cuda.empty_cache()
        d = enc()
        print(mode, "after NaN fill + empty_cache:", [cmp(x, y) for x, y in zip(a, d)], flush=True)
