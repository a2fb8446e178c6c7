"""diagnostic: where do form (i) and form (ii) of bench.extra_eval_object_loop differ when their uint8 images do?"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pixelnerf_amd import ops, util
from testdata import synthetic
dev = torch.device("cuda:0")
scene, meta, net, renderer, mlps = bench.build(dev, "f16x3", "sn64")
W, H = meta["W"], meta["H"]
rs = np.random.RandomState(7)
images = torch.from_numpy(rs.uniform(-1, 1, (4, 25, 3, H, W)).astype(np.float32))
src_pose = synthetic.pose_spherical(30.0, -20.0, meta["radius"])[None]
focal = torch.tensor(meta["focal"][0], dtype=torch.float32)[None]
c = torch.tensor(meta["c"], dtype=torch.float32)[None]
lats = []
with torch.no_grad():
    for rep in range(6):
        net.encode(images[rep % 2, :1].to(dev).unsqueeze(0), src_pose.to(dev).unsqueeze(0), focal.to(dev), c=c.to(dev))
        lats.append(net.encoder.latent.clone())
    for i in (2, 4):
        print("encode of image 0, call 0 vs call", i, "max abs diff", float((lats[0] - lats[i]).abs().max()), "equal", torch.equal(lats[0], lats[i]))
    print("encode of image 1, call 1 vs 3 / 5:", torch.equal(lats[1], lats[3]), torch.equal(lats[1], lats[5]))
    r = bench.extra_eval_object_loop(dev, n_views=24, n_obj=4)
    print({k: r[k] for k in r if "check" in k})
