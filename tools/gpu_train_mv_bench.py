#!/usr/bin/env python3
"""config-5 training step on the single-view scene and on the multi-view scene (2 objects x 2 source views), f16."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
dev = torch.device("cuda:0")
for scene in ("train", "train_mv", "train", "train_mv"):
    r = bench.extra_train_step(dev, "f16", scene, steps=40, warmup=8, with_graph=False)
    print(scene, "%.3f ms/step  %.1f TFLOP/s algorithmic  frac %.3f" % (r["ms_per_step"], r["algorithmic_tflops"], r["frac_of_f16_mfma_peak"]), flush=True)
