#!/usr/bin/env python3
"""BASELINE configs[4] training step at every training precision + eager PyTorch fp32 autograd on the same GPU."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")
res = {}
for prec, steps in (("f16", 30), ("f16x3", 10), ("f32", 4)):
    r = bench.extra_train_step(dev, prec, steps=steps, warmup=3, with_graph=False)
    res[prec] = {k: r[k] for k in ("ms_per_step", "rays_per_s", "algorithmic_tflops", "loss_first_step", "loss")}
    print(prec, json.dumps(res[prec]), flush=True)
if "--eager" in sys.argv:
    from oracle import pnr_oracle as O
    from testdata import synthetic
    O.USE_GRID_SAMPLE = True
    scene, meta = synthetic.make_scene("train")
    sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    sc["latent"] = sc["latent"].clone().requires_grad_(True)
    pc = {k: v.to(dev).requires_grad_(True) for k, v in synthetic.make_mlp_params(11).items()}
    pf = {k: v.to(dev).requires_grad_(True) for k, v in synthetic.make_mlp_params(12).items()}
    rays = synthetic.target_rays(meta, n_rays=128).to(dev)
    gt = torch.rand(4, 128, 3, device=dev)
    noise = {k: v.to(dev) for k, v in synthetic.make_noise(512, 64, 32, 16).items()}
    opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=1e-4)

    def step():
        out = O.render(sc, pc, pf, rays, noise, 64, 32, 16, white_bkgd=True)
        loss = ((out["coarse"]["rgb"] - gt) ** 2).mean() + ((out["fine"]["rgb"] - gt) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        sc["latent"].grad = None
        loss.backward()
        opt.step()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    res["eager_torch_fp32"] = {"ms_per_step": (time.perf_counter() - t0) / 5 * 1e3}
    print("eager", json.dumps(res["eager_torch_fp32"]), flush=True)
print(json.dumps(res))
