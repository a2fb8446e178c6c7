#!/usr/bin/env python3
"""Run selected bench extras on their own:  python tools/gpu_extras_quick.py eval_object_loop dtu_9v ..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
dev = torch.device("cuda:0")
fns = {"eval_object_loop": lambda: bench.extra_eval_object_loop(dev),
       "dtu_9v": lambda: bench.extra_render_config(dev, "dtu_9v", 1, n_oracle=32, n_f32=2048, steps=2, precisions=("f16x3",)),
       "dtu": lambda: bench.extra_render_config(dev, "dtu", 1),
       "srn_car": lambda: bench.extra_render_config(dev, "srn_car", 4),
       "train_step_fp32_class": lambda: bench.extra_train_step(dev, "f16x3", steps=16, warmup=4, with_graph=True)}
for k in sys.argv[1:]:
    print(k, json.dumps(fns[k](), indent=1))
