#!/usr/bin/env python3
"""fp32-class (f16x3) fused training step, config-5 scene: ms/step (for A/B of library variants via PIXELNERF_HIP_LIB)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
dev = torch.device("cuda:0")
bench.extra_train_step(dev, "f16", steps=10, warmup=3, with_graph=False)  # clocks up
for rnd in range(2):
    r = bench.extra_train_step(dev, "f16x3", sys.argv[1] if len(sys.argv) > 1 else "train", steps=10, warmup=3, with_graph=False)
    print("%.3f ms/step" % r["ms_per_step"], flush=True)
