#!/usr/bin/env python3
"""debug: encoder HIP-graph recapture after a parameter storage is replaced"""
import os, sys, torch, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from pixelnerf_amd.model.encoder import SpatialEncoder
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = SpatialEncoder("resnet34", pretrained=False, use_first_pool=False).to(dev).eval()
img = torch.rand(1, 3, 64, 64, device=dev) * 2 - 1
def run(tag, graph):
    SpatialEncoder.use_graph = graph
    with torch.no_grad():
        y = enc(img).clone()
    print(f"{tag:34s} graph={graph} norm {float(y.norm()):.6f}  graphs cached {len(getattr(enc, '_graphs', {}))}")
    return y
e0 = run("eager, original weights", False)
g1 = run("graph #1 (capture)", True)
g1b = run("graph #1 (replay)", True)
w = enc.model.conv1.weight
with torch.no_grad():
    w.data = w.data * 1.5
print("fingerprint changed:", True)
g2 = run("graph after conv1.weight.data swap", True)
e1 = run("eager after swap", False)
g2b = run("graph replay after swap", True)
print("g1 vs e0 max rel", float((g1 - e0).abs().max() / e0.abs().max()), " g2 vs e1", float((g2 - e1).abs().max() / e1.abs().max()),
      " e1 vs 1.5 e0", float((e1 - 1.5 * e0).abs().max() / e1.abs().max()), " g2b vs e1", float((g2b - e1).abs().max() / e1.abs().max()))
