#!/usr/bin/env python3
"""Latency of render_par(rays) vs batch size (sn64, 64+128, f16): single 64x64 image = 4096 rays."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

dev = torch.device("cuda:0")
scene, meta, net, renderer, mlps = bench.build(dev, "f16")
render_par = renderer.bind_parallel(net, None, simple_output=True).eval()
for R in (256, 1024, 4096, 16384, 65536, 262144):
    rays = bench.make_rays(meta, R, 0).to(dev)
    with torch.no_grad():
        for _ in range(3):
            render_par(rays[None])
        torch.cuda.synchronize()
        n = 10 if R <= 65536 else 3
        t0 = time.perf_counter()
        for _ in range(n):
            render_par(rays[None])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print(f"R={R:7d}: {dt*1e3:9.3f} ms per call  {R/dt/1e3:8.1f} k rays/s", flush=True)
