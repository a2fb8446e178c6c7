#!/usr/bin/env python3
"""TEST / ANALYSIS INFRASTRUCTURE (oracle side, CPU only; nothing in the product imports it).

How much accuracy the 512-wide linears keep under cheaper operand schemes than the shipped fp32-class one, measured on the CPU
restatement of the reference (pnr_oracle.pixelnerf_forward) with the operands of every blocks[b].fc_0 / fc_1 product quantised
and the accumulation left in fp64 -- i.e. the operand error alone:

    f16      w, x rounded to fp16                              (precision="f16", 1 MFMA per product)
    f16x3    (head, tail) fp16 pairs, wh xh + wh xl + wl xh    (precision="f16x3", 3 MFMAs: shipped default)
    f8tail   head fp16; the two tail products with fp8 (e4m3) operands   (gfx950: 2x the f16 MFMA rate -> 2 "units" instead of 3)
    f6tail   ... with block-scaled fp6 (e2m3, MX) operands               (4x the rate -> 1.5 units)

Result (python oracle/sim_tail_precision.py sn64 srn_car; 6144 points each): per-point max |rgb| error vs fp64
    f16 7.2e-4 / 5.7e-4,  f16x3 9.8e-7 / 8.5e-7,  f8tail 2.9e-5 / 1.9e-5,  f6tail 3.2e-5 / 2.0e-5.
The 8- and 6-bit tail schemes are 25x tighter than f16 and 30x looser than f16x3: not the reference's own arithmetic class,
which is why docs/HISTORY.md section 8 lists them as considered and not built.
"""
import sys, torch, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import pnr_oracle as O
from testdata import synthetic as S
torch.manual_seed(0)
F8=torch.float8_e4m3fn
def q16(t): return t.to(torch.float16).to(torch.float64)
def q8(t, s):   # fp8 e4m3 with power-of-two scale s
    return (t*s).to(torch.float32).to(F8).to(torch.float64)/s
def q6(t, blk=32):  # e2m3 block-scaled (MX) along last dim, 32-element blocks
    sh=t.shape; x=t.reshape(-1, sh[-1]//blk, blk)
    m=x.abs().amax(-1,keepdim=True).clamp_min(1e-300)
    e=torch.floor(torch.log2(m))-2   # scale so max in [4,8) -> e2m3 max 7.5
    y=x/2.0**e
    a=y.abs(); sgn=torch.sign(y)
    # e2m3: normals 1..7.5 with 3 mantissa bits, subnormal step 0.125
    ex=torch.floor(torch.log2(a.clamp_min(1e-300))).clamp(0,2)
    step=2.0**(ex-3)
    qv=torch.round(a/step)*step
    qv=qv.clamp(max=7.5)
    return (sgn*qv*2.0**e).reshape(sh)
MODE='exact'
def lin512(x, W, b):
    if MODE=='exact': return torch.nn.functional.linear(x,W,b)
    wh=q16(W); xh=q16(x)
    if MODE=='f16': return torch.nn.functional.linear(xh,wh,b)
    wl=q16(W-wh); xl=q16(x-xh)
    L=torch.nn.functional.linear
    if MODE=='f16x3': return L(xh,wh,b)+L(xl,wh)+L(xh,wl)
    if MODE=='f8tail':
        return L(xh,wh,b)+L(q8(xl,2.0**12),q8(wh,2.0**6))+L(q8(xh,1.0),q8(wl,2.0**18))
    if MODE=='f8tail_rawx':   # fp8 of full x / w rather than of heads (same thing basically)
        return L(xh,wh,b)+L(q8(xl,2.0**12),q8(W,2.0**6))+L(q8(x,1.0),q8(wl,2.0**18))
    if MODE=='f6tail':
        return L(xh,wh,b)+L(q6(xl),q6(wh))+L(q6(xh),q6(wl))
    if MODE=='f8x_f6w':
        return L(xh,wh,b)+L(q8(xl,2.0**12),q6(wh))+L(q8(xh,1.0),q6(wl))
    raise ValueError
def rf(p, zx, combine_inner_dims, d_latent=512, n_blocks=5, combine_layer=3, return_hidden=False):
    z=zx[...,:d_latent]; x=zx[...,d_latent:]
    L=torch.nn.functional.linear
    x=L(x,p["lin_in.weight"],p["lin_in.bias"])
    for b in range(n_blocks):
        if b==combine_layer and not (len(combine_inner_dims)==1 and combine_inner_dims[0]==1):
            x=x.reshape(-1,*combine_inner_dims,*x.shape[1:]).mean(dim=1)
        if b<combine_layer:
            x=x+L(z,p[f"lin_z.{b}.weight"],p[f"lin_z.{b}.bias"])
        net=lin512(torch.relu(x),p[f"blocks.{b}.fc_0.weight"],p[f"blocks.{b}.fc_0.bias"])
        dx=lin512(torch.relu(net),p[f"blocks.{b}.fc_1.weight"],p[f"blocks.{b}.fc_1.bias"])
        x=x+dx
    out=L(torch.relu(x),p["lin_out.weight"],p["lin_out.bias"])
    return (out,x) if return_hidden else out
O.resnetfc_forward=rf

def run(name, R=96):
    global MODE
    scene, meta = S.make_scene(name)
    mlp = {k: v.double() for k, v in S.make_mlp_params(11).items()}
    sc = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in scene.items()}
    rays = S.target_rays(meta, n_rays=R)[:, :R].double()
    SB = rays.shape[0]
    t = torch.rand(SB, R, 64, dtype=torch.float64)
    z = rays[..., 6:7] * (1 - t) + rays[..., 7:8] * t
    xyz = (rays[..., None, :3] + z[..., None] * rays[..., None, 3:6]).reshape(SB, -1, 3)
    vd = rays[..., None, 3:6].expand(-1, -1, 64, -1).reshape(SB, -1, 3)
    res = {}
    for m in ['exact', 'f16', 'f16x3', 'f8tail', 'f6tail', 'f8x_f6w']:
        MODE = m
        res[m] = O.pixelnerf_forward(sc, mlp, xyz, vd)
    ex = res['exact']
    print(f"{name}: {ex.shape[0]*ex.shape[1]} points; max |rgb| err, rms rgb err, max sigma rel err")
    for m in res:
        if m == 'exact': continue
        d = res[m] - ex
        print(f"  {m:10s} rgb max {d[..., :3].abs().max():.3e} rms {d[..., :3].pow(2).mean().sqrt():.3e}  sigma max {d[..., 3].abs().max():.3e} (sigma scale {ex[..., 3].abs().max():.2f})")
for n in sys.argv[1:] or ['sn64']:
    run(n)
