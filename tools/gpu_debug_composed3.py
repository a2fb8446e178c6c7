#!/usr/bin/env python3
"""debug: every linear_backward call of the failing composed case, f16x3 against f32 ON THE SAME INPUTS"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from testdata import synthetic
from pixelnerf_amd.model.resnetfc import ResnetFC
from pixelnerf_amd import ops
dev = torch.device("cuda:0")
def rel(a, b):
    if a is None: return float("nan")
    a, b = a.detach().double().reshape(-1), b.detach().double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))
d_in, d_latent, kw, (G, NS, B) = 42, 512, dict(d_hidden=512, n_blocks=5, combine_layer=3), (1, 2, 40)
orig = ops.linear_backward
n = [0]
def wrapped(dy, x, weight, relu_in=False, need_dx=True, need_dw=True, need_db=True, precision="f16x3"):
    a = orig(dy, x, weight, relu_in=relu_in, need_dx=need_dx, need_dw=need_dw, need_db=need_db, precision="f16x3")
    b = orig(dy, x, weight, relu_in=relu_in, need_dx=need_dx, need_dw=need_dw, need_db=need_db, precision="f32")
    ref = dy.reshape(-1, weight.shape[0]).double() @ weight.double()
    if relu_in: ref = ref * (x.reshape(-1, weight.shape[1]) > 0)
    print(f"call {n[0]:2d}: rows {dy.reshape(-1, weight.shape[0]).shape[0]} {tuple(weight.shape)} relu {relu_in}: dx {rel(a[0], b[0]):.1e} (f32 vs fp64 torch {rel(b[0], ref) if b[0] is not None else float('nan'):.1e}, f16x3 vs fp64 {rel(a[0], ref) if a[0] is not None else float('nan'):.1e}) dw {rel(a[1], b[1]):.1e} db {rel(a[2], b[2]):.1e}"
          f"  dy strides {dy.stride()} contiguous {dy.is_contiguous()}")
    n[0] += 1
    return b
ops.linear_backward = wrapped
mlp = ResnetFC(d_in, d_latent=d_latent, **kw)
shapes = [(k, tuple(v.shape)) for k, v in mlp.state_dict().items()]
mlp.load_state_dict(synthetic.fill_state(shapes, 5)); mlp = mlp.to(dev)
g = torch.Generator().manual_seed(1)
zx = (torch.randn(G * NS * B, d_latent + d_in, generator=g) * 0.7).to(dev).requires_grad_(True)
out = mlp(zx, combine_inner_dims=(NS, B))
w_out = torch.randn(out.shape, generator=g).to(dev)
(out * w_out).sum().backward()
