#!/usr/bin/env python3
"""fp32-class (f16x3) training step, BASELINE configs[4] + the multi-view scene: the fused path (default) against the GEMM-per-layer
form (autograd.FUSED_SPLIT_TRAINING = False), same process, alternating."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pixelnerf_amd import autograd  # noqa: E402

dev = torch.device("cuda:0")
bench.extra_train_step(dev, "f16", steps=10, warmup=3, with_graph=False)  # clocks up
for scene in ("train", "train_mv"):
    for rnd in range(2):
        for fused in (True, False):
            autograd.FUSED_SPLIT_TRAINING = fused
            r = bench.extra_train_step(dev, "f16x3", scene, steps=8, warmup=2, with_graph=False)
            print(scene, "fused         " if fused else "GEMM per layer", "%.3f ms/step  loss %.5f -> %.5f" % (r["ms_per_step"], r["loss_first_step"], r["loss"]), flush=True)
