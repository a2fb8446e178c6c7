"""diagnostic: eval_object_loop's identity check with float outputs, latents and tables compared, after other extras ran in the process"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pixelnerf_amd import ops, util
from pixelnerf_amd.util import DotMap
from testdata import synthetic
dev = torch.device("cuda:0")
pre = sys.argv[1:] 
for name in pre:
    if name == "train": bench.extra_train_step(dev, "f16x3", steps=4, warmup=2, with_graph=False)
    elif name == "f32": bench.extra_train_step(dev, "f32", steps=2, warmup=1, with_graph=False)
    else: bench.extra_render_config(dev, name, 1, n_oracle=16, n_f32=1024, steps=1, precisions=("f16x3",))
    print("ran", name, flush=True)
scene, meta, net, renderer, mlps = bench.build(dev, "f16x3", "sn64")
W, H, P = meta["W"], meta["H"], meta["W"] * meta["H"]
z_near, z_far, focal_xy, c_xy = meta["z_near"], meta["z_far"], meta["focal"], meta["c"]
n_views = 24
rs = np.random.RandomState(7)
images = torch.from_numpy(rs.uniform(-1, 1, (4, 1 + n_views, 3, H, W)).astype(np.float32))
src_pose = synthetic.pose_spherical(30.0, -20.0, meta["radius"])[None]
tgt_poses = torch.stack([synthetic.pose_spherical(45.0 + 13.0 * i, -20.0, meta["radius"]) for i in range(n_views)])
focal = torch.tensor(focal_xy[0], dtype=torch.float32)[None]
c = torch.tensor(c_xy, dtype=torch.float32)[None]
render_par = renderer.bind_parallel(net, None, simple_output=False).eval()
key = 0x1234567
def enc(o):
    net.encode(images[o, :1].to(dev).unsqueeze(0), src_pose.to(dev).unsqueeze(0), focal.to(dev), c=c.to(dev))
    return net.encoder.latent.clone(), net.tables(True).clone(), net.tables(False).clone()
with torch.no_grad():
    for trial in range(3):
        la, tca, tfa = enc(1)
        out = ops.render_views(net.scene(), net.packed(True), net.packed(False), tgt_poses.to(dev), W, H, focal_xy, z_near, z_far, 64, 128, 16, None, c=c_xy,
                               white_bkgd=meta["white_bkgd"], tables=(net.tables(True), net.tables(False)), seed=key)
        b_c, b_f, b_d = out["coarse"]["rgb"].reshape(-1, 3).clone(), out["fine"]["rgb"].reshape(-1, 3).clone(), out["fine"]["depth"].reshape(-1).clone()
        lb, tcb, tfb = enc(1)
        all_rays = util.gen_rays(tgt_poses.to(dev), W, H, focal, z_near, z_far, c=c).reshape(-1, 8)
        a_c, a_f, a_d, lo = [], [], [], 0
        for rays in torch.split(all_rays, 50000, dim=0):
            renderer.ray_id_offset, renderer.ray_id_stride, renderer._seed_override = lo, all_rays.shape[0], key
            r = DotMap(render_par(rays[None]))
            a_c.append(r.coarse.rgb[0]); a_f.append(r.fine.rgb[0]); a_d.append(r.fine.depth[0]); lo += rays.shape[0]
        renderer.ray_id_offset, renderer.ray_id_stride, renderer._seed_override = 0, 0, None
        a_c, a_f, a_d = torch.cat(a_c), torch.cat(a_f), torch.cat(a_d)
        # whole set in one call through the renderer as well
        renderer.ray_id_offset, renderer.ray_id_stride, renderer._seed_override = 0, all_rays.shape[0], key
        w = DotMap(render_par(all_rays[None])); renderer.ray_id_offset, renderer.ray_id_stride, renderer._seed_override = 0, 0, None
        def cmp(x, y): return "equal" if torch.equal(x, y) else "max|d| %.3e in %d of %d" % (float((x - y).abs().max()), int((x != y).sum()), x.numel())
        print(f"trial {trial}: latent {cmp(la, lb)}; tables c {cmp(tca, tcb)} f {cmp(tfa, tfb)}; coarse rgb {cmp(a_c, b_c)}; fine rgb {cmp(a_f, b_f)}; depth {cmp(a_d, b_d)}; "
              f"whole-call-through-renderer vs render_views: coarse {cmp(w.coarse.rgb[0], b_c)} fine {cmp(w.fine.rgb[0], b_f)}", flush=True)
