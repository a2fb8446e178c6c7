"""
GPU tests (-m gpu) of the training path (BASELINE config 5): gradients from the HIP backward
(pixelnerf_amd.autograd: compositing backward, fused data-gradient chain, latent scatter-add,
library GEMMs for dW) against torch autograd through the CPU oracle.

Reference semantics: train/train.py:199-215 back-propagates MSE(coarse rgb) + MSE(fine rgb)
through NeRFRenderer.forward into both ResnetFCs and encoder.latent, including the position
gradient through the n_fine_depth samples (nerf.py:292: the coarse depth is NOT detached).  The
comparison is against the oracle with exactly those semantics; tests/gpu_grad_check.py also
prints how large that position term is (profiles/r01_grad_parity_table.txt).

Tolerances (16-bit MFMA operands, fp32 accumulation, fp32 library GEMMs for dW):
  f16 : per-tensor relative L2 error <= 3e-2, cosine >= 0.9995
  bf16: per-tensor relative L2 error <= 8e-2, cosine >= 0.997
Compositing backward alone is fp32 on both sides: 2e-5 relative.
"""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import golden_setup
from oracle import pnr_oracle as O

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # gpu_grad_check.py (the gradient parity table, also a script) lives next to the tests

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ["sn64_64_128", "dtu_mini_64_128", "mv_mini_lindisp"])
def test_composite_backward_matches_autograd(dev, name):
    from pixelnerf_amd import ops
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    r = rays.reshape(-1, 8)
    white = bool(g["white_bkgd"])
    z = torch.from_numpy(g["fine_z"])
    out = torch.from_numpy(g["fine_rgbsigma"]).clone().requires_grad_(True)
    w, rgb, depth = O.composite_from_rgbsigma(r, z, out, white)
    gen = torch.Generator().manual_seed(9)
    d_rgb, d_depth, d_w = torch.randn(rgb.shape, generator=gen), torch.randn(depth.shape, generator=gen), \
        torch.randn(w.shape, generator=gen)
    (rgb * d_rgb).sum().backward(retain_graph=True)
    g_rgb_only = out.grad.clone()
    out.grad = None
    ((rgb * d_rgb).sum() + (depth * d_depth).sum() + (w * d_w).sum()).backward()
    g_all = out.grad
    a = ops.composite_backward(r.to(dev), z.to(dev), out.detach().to(dev), white, d_rgb.to(dev)).cpu()
    b = ops.composite_backward(r.to(dev), z.to(dev), out.detach().to(dev), white, d_rgb.to(dev), d_depth.to(dev),
                               d_w.to(dev)).cpu()
    for got, ref in ((a, g_rgb_only), (b, g_all)):
        rel = (got - ref).norm() / ref.norm()
        assert rel <= 2e-5, rel
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))


@pytest.mark.parametrize("prec,rel_tol,cos_tol", [("f16", 3e-2, 0.9995), ("bf16", 8e-2, 0.997)])
@pytest.mark.parametrize("name", ["train_64_32", "srn_mini_64_128"])  # SB=4 x NS=1 (config 5 shapes); NS=2 pooling
def test_parameter_and_latent_gradients_match_oracle_autograd(dev, name, prec, rel_tol, cos_tol):
    import gpu_grad_check
    rows, (loss_o, loss_h) = gpu_grad_check.compare(name, prec, verbose=False)
    assert abs(loss_o - loss_h) <= 2e-3 * abs(loss_o)
    assert len(rows) == 1 + 2 * 30
    for k, norm, rel, cos, _ in rows:
        assert norm > 0, k
        assert rel <= rel_tol, f"{k}: rel err {rel:.3e}"
        assert cos >= cos_tol, f"{k}: cos {cos:.6f}"


def test_training_step_updates_and_is_deterministic(dev):
    """One optimiser step through the reference-style loop: render_par(rays, want_weights=True)
    -> MSE coarse + fine -> backward -> Adam (train/train.py:199-215, trainlib/trainer.py:232-237).
    Parameter gradients are bit-reproducible (fixed-order reductions, no atomics); so is the latent gradient on grids that take
    the LDS-slab scatter (one zeroed buffer per pass, at most two commuting adds per element: pnr_bwd.hip, autograd.py)."""
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util import DotMap
    from pixelnerf_amd.util.conf import default_model_conf
    from helpers import mlp_params
    g, scene, meta, mc, mf, rays, noise = golden_setup("train_64_32")
    net = make_model(default_model_conf(), precision="f16").to(dev).train()
    net.mlp_coarse.load_state_dict(mlp_params(11))
    net.mlp_fine.load_state_dict(mlp_params(12))
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([32.0, 32.0], device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).to(dev)
    render_par = rend.bind_parallel(net, None, simple_output=False).train()
    params = list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters())
    opt = torch.optim.Adam(params, lr=1e-4)
    gt = torch.rand(4, 32, 3, device=dev)
    r = rays.to(dev)

    def step():
        torch.manual_seed(3)
        rd = DotMap(render_par(r, want_weights=True))
        assert len(rd.fine) > 0 and rd.coarse.rgb.requires_grad
        loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
        opt.zero_grad()
        lat.grad = None
        loss.backward()
        return loss.item(), [p.grad.clone() for p in params], lat.grad.clone()

    l1, g1, gl1 = step()
    l2, g2, gl2 = step()
    assert l1 == l2 and all(torch.equal(a, b) for a, b in zip(g1, g2))
    assert torch.equal(gl1, gl2)
    before = [p.detach().clone() for p in params]
    opt.step()
    assert all(not torch.equal(a, p.detach()) for a, p in zip(before, params))
    l3, _, _ = step()  # weights changed -> repacked streams -> different loss
    assert l3 != l1
    # stop_encoder_grad (train/train.py:65): no latent gradient
    net.stop_encoder_grad = True
    rd = DotMap(render_par(r, want_weights=True))
    lat.grad = None
    ((rd.fine.rgb - gt) ** 2).mean().backward()
    assert lat.grad is None


def test_training_converges_on_a_fixed_batch(dev):
    """Normalised gradient descent on one fixed ray batch with frozen noise (a smooth deterministic
    loss): each step is sized to predict a 10 % decrease (eta = 0.1 L / |g|^2), so the loss must
    fall monotonically if -- and only if -- the HIP gradients (both MLPs + encoder.latent) point
    downhill with the right scale."""
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util import DotMap
    from pixelnerf_amd.util.conf import default_model_conf
    from helpers import mlp_params
    g, scene, meta, mc, mf, rays, noise = golden_setup("train_64_32")
    net = make_model(default_model_conf(), precision="f16").to(dev).train()
    net.mlp_coarse.load_state_dict(mlp_params(11))
    net.mlp_fine.load_state_dict(mlp_params(12))
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    ls = torch.tensor([32.0, 32.0], device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).to(dev)
    render_par = rend.bind_parallel(net, None, simple_output=False).train()
    params = list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters()) + [lat]
    gt = torch.rand(4, 32, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) * 0.5 + 0.25
    r = rays.to(dev)
    losses = []
    for it in range(9):
        net.encoder.latent = lat  # leaf with grad, standing in for the encoder output
        torch.manual_seed(123)    # frozen noise -> the same sample positions every step
        rd = DotMap(render_par(r, want_weights=True))
        loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
        for p_ in params:
            p_.grad = None
        loss.backward()
        losses.append(loss.item())
        g2 = sum(float((p_.grad.double() ** 2).sum()) for p_ in params)
        assert g2 > 0 and all(torch.isfinite(p_.grad).all() for p_ in params)
        eta = 0.1 * loss.item() / g2
        with torch.no_grad():
            for p_ in params:
                p_.add_(p_.grad, alpha=-eta)  # in-place: bumps _version -> weights are re-packed
    assert all(b < a for a, b in zip(losses, losses[1:])), losses
    assert losses[-1] < 0.6 * losses[0], losses


# LDS-slab kernel (16 / 8 / 4 channels with padded rows, 4 channels unpadded: 64x64; one owner per (image, slice) or -- 600 rays x 20
# samples on 4 x 32 slices -- two workgroups per pair that meet in HBM with atomics; K = 10 / 20: ray boundaries off the segment
# grid) / large grids (5760 texels = 3 x 3 tiles with ragged edges; the DTU grid 150 x 200; 33 x 97: a tile row of ONE texel row): the
# tiled form -- a workgroup per (image, 32 x 32-texel tile, 16-channel slice), segments listed in every tile their corners touch
@pytest.mark.parametrize("Hl,Wl,n_rays,K", [(16, 16, 24, 10), (40, 40, 24, 10), (50, 60, 24, 10), (64, 64, 24, 10), (72, 80, 24, 10),
                                           (16, 16, 600, 20), (32, 32, 128, 96), (150, 200, 64, 24), (33, 97, 200, 16)])
def test_latent_scatter_matches_autograd(dev, Hl, Wl, n_rays, K):
    """d(interpolated latent) -> d(feature grid) for SB=2 x NS=2, against autograd through the oracle's lookup
    (encoder.py:80-109).  fp32 on both sides (the slab kernel sums per-segment fp32 partial sums in fp64); 1e-5 relative."""
    from helpers import scene_for
    from pixelnerf_amd import ops
    from testdata import synthetic
    scene, meta = scene_for("mv_mini")
    scene = dict(scene)
    gen = torch.Generator().manual_seed(21)
    scene["latent"] = torch.randn(4, 512, Hl, Wl, generator=gen)
    SB, NS = scene["SB"], scene["NS"]
    rays = synthetic.target_rays(meta, n_rays=n_rays)  # (2, n_rays, 8)
    r = rays.reshape(-1, 8)
    z = O.sample_coarse(r, torch.rand(r.shape[0], K, generator=gen), K)
    B = n_rays * K
    P = SB * B
    d_zlat = torch.randn(NS * P, 512, generator=gen)
    d_zlat[5] = 0.0
    # reference: the oracle's projection + lookup, differentiated w.r.t. the grid
    lat = scene["latent"].clone().requires_grad_(True)
    xyz = (r[:, None, :3] + z.unsqueeze(2) * r[:, None, 3:6]).reshape(SB, B, 3)
    xyz_r = O.repeat_interleave(xyz, NS)
    poses = scene["poses"]
    xyz_cam = torch.matmul(poses[:, None, :3, :3], xyz_r.unsqueeze(-1))[..., 0] + poses[:, None, :3, 3]
    uv = -xyz_cam[:, :, :2] / xyz_cam[:, :, 2:]
    uv = uv * scene["focal"].unsqueeze(1) + scene["c"].unsqueeze(1)
    feats = O.index_latent(lat, uv, scene["image_shape"])  # (SB*NS, 512, B)
    g = d_zlat.reshape(NS, SB, B, 512).permute(1, 0, 3, 2).reshape(SB * NS, 512, B)  # row obj*NS+view
    (feats * g).sum().backward()
    sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev),
                        scene["image_shape"], NS)
    out = torch.zeros(4, Hl, Wl, 512, device=dev)
    ops.latent_scatter(sc, r.to(dev), z.to(dev), d_zlat.to(dev), out)
    ops.latent_scatter(sc, r.to(dev), z.to(dev), d_zlat.to(dev), out)  # accumulates into the buffer
    got = out.permute(0, 3, 1, 2).cpu() / 2
    ref = lat.grad
    assert (got - ref).norm() <= 1e-5 * ref.norm()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=2e-5 * float(ref.abs().max()))


def test_weight_grad_kernel_split_operands(dev):
    """dW = dY^T X at split-operand precision (pnr_weight_grad_batched at PNR_PREC_F16X3, dw_split_kernel): operands as
    [head | tail] f16 row sets, three MFMAs per product.  Against an fp64 matmul of the fp32 values the pairs stand for:
    only the tail x tail term (2^-22 of a product) and the summation order differ.  Ragged rows, both storage orders, the
    narrow lin_in operand, several jobs in one launch, bit-reproducibility."""
    from pixelnerf_amd import _lib, ops
    gen = torch.Generator().manual_seed(9)
    perm = ops.storage_perm(dev).long()
    p = _lib.PREC_F16X3

    def pair(v):  # fp32 -> [head | tail] (2, rows, cols) f16, and the fp32 value the pair stands for
        h = v.to(torch.float16)
        l = (v - h.float()).to(torch.float16)
        return torch.stack([h, l]).contiguous(), h.double() + l.double()

    jobs, refs = [], []
    for rows, cols, rs, cs in ((1000, 512, True, True), (37, 512, False, False), (4096 + 5, 512, True, False), (3000, 64, True, False)):
        dYp, dYv = pair((torch.randn(rows, 512, generator=gen) * 0.5).to(dev))
        Xv32 = torch.randn(rows, cols, generator=gen).to(dev)
        if cols == 64:
            Xv32[:, 42:] = 0
        Xp, Xv = pair(Xv32)
        jobs.append((dYp, Xp, rs, cs, cols, 42 if cols == 64 else 512))
        dW, db = dYv.t() @ Xv * 0.25, dYv.sum(0) * 0.25
        if cols == 64:
            dW = dW[:, :42]
        if rs:
            o = torch.empty_like(dW); o[perm] = dW; dW = o
            b = torch.empty_like(db); b[perm] = db; db = b
        if cs:
            o = torch.empty_like(dW); o[:, perm] = dW; dW = o
        refs.append((dW, db))
    outs = ops.weight_grad_batched(jobs, p, 0.25)
    again = ops.weight_grad_batched(jobs, p, 0.25)
    for (dW, db), (dW2, db2), (rW, rb) in zip(outs, again, refs):
        assert dW.shape == rW.shape
        assert (dW.double() - rW).norm() <= 1e-6 * rW.norm() and (db.double() - rb).norm() <= 1e-6 * rb.norm()
        assert (dW.double() - rW).abs().max() <= 2e-6 * rW.abs().max()
        assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("prec,dt", [("f16", torch.float16), ("bf16", torch.bfloat16)])
def test_weight_grad_kernel_single_and_batched(dev, prec, dt):
    """dW = dY^T X / db = column sums from 16-bit row-major dumps (transposing LDS reads, split rows, fixed-order
    reduce) against an fp32 matmul of the same 16-bit values: products are exact in fp32, only the summation
    order differs -> 2e-6 relative.  Ragged row counts, storage-order -> feature-order mapping, several jobs
    with different row counts in one launch, bit-reproducibility."""
    from pixelnerf_amd import _lib, ops
    gen = torch.Generator().manual_seed(4)
    perm = ops.storage_perm(dev).long()  # storage position e -> feature
    p = _lib.PRECISIONS[prec]

    def ref(dY, X, rs, cs, scale):
        dW = (dY.float().t() @ X.float()) * scale
        db = dY.float().sum(0) * scale
        if rs:
            o = torch.empty_like(dW); o[perm] = dW; dW = o
            b = torch.empty_like(db); b[perm] = db; db = b
        if cs:
            o = torch.empty_like(dW); o[:, perm] = dW; dW = o
        return dW, db

    jobs = []
    for rows, rs, cs in ((1000, True, True), (37, False, False), (4096 + 5, True, False), (640, False, True)):
        dY = (torch.randn(rows, 512, generator=gen) * 0.5).to(dt).to(dev)
        X = torch.randn(rows, 512, generator=gen).to(dt).to(dev)
        jobs.append((dY, X, rs, cs))
    outs = ops.weight_grad_batched(jobs, p, 0.25)
    again = ops.weight_grad_batched(jobs, p, 0.25)
    for (dY, X, rs, cs), (dW, db), (dW2, db2) in zip(jobs, outs, again):
        rW, rb = ref(dY, X, rs, cs, 0.25)
        assert (dW - rW).norm() <= 2e-6 * rW.norm() and (db - rb).norm() <= 2e-6 * rb.norm()
        assert (dW - rW).abs().max() <= 1e-4 * rW.abs().max()
        assert torch.equal(dW, dW2) and torch.equal(db, db2)
        sW, sb = ops.weight_grad(dY, X, p, 0.25, rows_st=rs, cols_st=cs)  # single-job entry: same maths
        assert (sW - rW).norm() <= 2e-6 * rW.norm() and (sb - rb).norm() <= 2e-6 * rb.norm()
    with pytest.raises(_lib.PixelNerfHipError):
        ops.weight_grad_batched(jobs * 5, p)  # more than 16 jobs
    # the lin_in form: X is the (rows,64) code|viewdir operand in natural order, dW is (512,42)
    rows = 3000
    dY = (torch.randn(rows, 512, generator=gen) * 0.5).to(dt).to(dev)
    X = torch.randn(rows, 64, generator=gen).to(dt).to(dev)
    X[:, 42:] = 0
    (dW, db), (dW_full, _) = ops.weight_grad_batched([(dY, X, True, False, 64, 42), jobs[0]], p, 2.0)
    rW = torch.empty(512, 42, device=dev)
    rW[perm] = (dY.float().t() @ X.float()[:, :42]) * 2.0
    rb = torch.empty(512, device=dev)
    rb[perm] = dY.float().sum(0) * 2.0
    assert dW.shape == (512, 42) and (dW - rW).norm() <= 2e-6 * rW.norm() and (db - rb).norm() <= 2e-6 * rb.norm()
    assert (dW_full - ref(*jobs[0], 2.0)[0]).norm() <= 2e-6 * dW_full.norm()
    # lin_out: g (P,4) fp32 x dump (P,512) in storage order -> (4,512) in feature order
    P = 5000
    g = torch.randn(P, 4, generator=gen).to(dev)
    x5 = torch.randn(P, 512, generator=gen).abs().to(dt).to(dev)
    oW, ob = ops.lin_out_grad(g, x5, p)
    r = torch.empty(4, 512, device=dev)
    r[:, perm] = g.t() @ x5.float()
    assert (oW - r).norm() <= 2e-6 * r.norm() and torch.allclose(ob, g.sum(0), rtol=1e-5, atol=1e-4)
    oW2, ob2 = ops.lin_out_grad(g, x5, p)
    assert torch.equal(oW, oW2) and torch.equal(ob, ob2)


def test_grad_scale_is_picked_on_device(dev):
    """scale = 2^(6 - ceil(log2 max|g|)) without a host sync; zero gradient -> 1; non-finite -> NaN poison.
    A device-side scale gives the same backward dumps as the same scale passed from the host."""
    import math
    from pixelnerf_amd import ops
    gen = torch.Generator().manual_seed(3)
    for mag in (3e-7, 0.02, 1.0, 64.0, 5000.0):
        g = (torch.randn(1000, 4, generator=gen) * mag).to(dev)
        sc = ops.grad_scale(g).cpu()
        want = 2.0 ** (6 - math.ceil(math.log2(float(g.abs().max()))))
        assert sc[0].item() == want and sc[1].item() == 1.0 / want
    assert ops.grad_scale(torch.zeros(8, 4, device=dev)).cpu().tolist() == [1.0, 1.0]
    bad = torch.ones(8, 4, device=dev)
    bad[3, 1] = float("inf")
    assert torch.isnan(ops.grad_scale(bad)).all()
    # host float vs device scalar: identical launches
    g_, scene, meta, mc, mf, rays, noise = golden_setup("train_64_32")
    sc_ = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev),
                         scene["image_shape"], scene["NS"])
    params = {k: v.to(dev) for k, v in mf.items()}
    pk, pkb = ops.pack_mlp(params, "f16"), ops.pack_mlp(params, "f16", backward=True)
    r = rays.reshape(-1, 8).to(dev)
    z = torch.from_numpy(g_["coarse_z"]).to(dev)
    _, dumps = ops.eval_ray_samples_train(sc_, pk, r, z)
    g_out = torch.randn(r.shape[0] * z.shape[1], 4, generator=gen).to(dev) * 0.01
    s = ops.grad_scale(g_out)
    a = ops.mlp_backward(pkb, dumps, g_out, float(s[0].item()))
    b = ops.mlp_backward(pkb, dumps, g_out, s[0:1])
    assert torch.equal(a.g_x0, b.g_x0) and all(torch.equal(x, y) for x, y in zip(a.g_fc0, b.g_fc0))


@pytest.mark.parametrize("name,prec", [("train_64_32", "f16"), ("srn_mini_64_128", "f16"), ("srn_mini_64_128", "bf16")])
def test_fused_chain_latent_and_input_gradients(dev, name, prec):
    """d z_lat = sum_b dY_b W_z[b] and d(code | viewdir) = dY_0 W_in, computed inside pnr_mlp_backward (transposed weight
    streams), against the same products formed in fp32 from the chain's own 16-bit dY dumps and the 16-bit-rounded weights:
    identical operands, fp32 accumulation on both sides -> agreement at accumulation-order level."""
    from pixelnerf_amd import ops
    g_, scene, meta, mc, mf, rays, noise = golden_setup(name)
    sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev),
                        scene["image_shape"], scene["NS"])
    params = {k: v.to(dev) for k, v in mf.items()}
    pk, pkb = ops.pack_mlp(params, prec), ops.pack_mlp(params, prec, backward=True)
    r = rays.reshape(-1, 8).to(dev)
    z = torch.from_numpy(g_["coarse_z"]).to(dev)[:, :37].contiguous()  # 37 samples per ray: ragged last tile
    _, dumps = ops.eval_ray_samples_train(sc, pk, r, z)
    gen = torch.Generator().manual_seed(5)
    g_out = (torch.randn(r.shape[0] * z.shape[1], 4, generator=gen) * 0.02).to(dev)
    s = ops.grad_scale(g_out)
    bd = ops.mlp_backward(pkb, dumps, g_out, s[0:1])
    torch.cuda.synchronize()
    dt = torch.float16 if prec == "f16" else torch.bfloat16
    perm = ops.storage_perm(dev)  # perm[e] = feature at storage position e
    inv_s = float(s[1].item())
    gz = [bd.g_x0, bd.g_fc1[0], bd.g_fc1[1]]
    want = sum(gz[b].float() @ params[f"lin_z.{b}.weight"][perm].to(dt).float() for b in range(3)) * inv_s
    want_in = (bd.g_x0.float() @ params["lin_in.weight"][perm].to(dt).float()) * inv_s
    rows = scene["NS"] * r.shape[0] * z.shape[1]
    assert bd.d_zlat.shape == (rows, 512) and bd.d_in.shape == (rows, 42)
    for got, ref, what in ((bd.d_zlat, want, "d_zlat"), (bd.d_in, want_in, "d_in")):
        assert torch.isfinite(got).all()
        err = (got - ref).abs().max().item()
        assert err <= 2e-5 * max(ref.abs().max().item(), 1e-12) + 1e-12, f"{what}: max err {err:.3e} vs scale {ref.abs().max().item():.3e}"


@pytest.mark.parametrize("name", ["train_64_32", "srn_mini_64_128", "dtu_mini_64_128"])  # NS = 1 (SB = 4), 2, 3
def test_hip_gradients_match_reference_autograd_goldens(dev, name):
    """HIP training path vs the gradients of the UNMODIFIED reference's own backward (tests/golden/gradients.npz,
    frozen by oracle/make_goldens.py): loss to 2e-3 relative, every one of the 61 gradient tensors within 3e-2
    relative L2 on its frozen subsample and within 3e-2 on its norm (f16 operands, fp32 accumulation)."""
    import gpu_grad_check
    from helpers import load_golden
    from testdata import synthetic
    gg = load_golden("gradients")
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    gt = torch.from_numpy(gg[f"{name}_gt"])

    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util.conf import default_model_conf
    net = make_model(default_model_conf(), precision="f16").to(dev).train()
    net.mlp_coarse.load_state_dict(mc)
    net.mlp_fine.load_state_dict(mf)
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32, device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, white_bkgd=bool(g["white_bkgd"]),
                        lindisp=bool(g["lindisp"])).to(dev).train()
    out = rend(net, rays.to(dev), want_weights=True, _noise={k: v.to(dev) for k, v in noise.items()})
    loss = ((out.coarse.rgb - gt.to(dev)) ** 2).mean() + ((out.fine.rgb - gt.to(dev)) ** 2).mean()
    loss.backward()
    ref_loss = float(gg[f"{name}_loss"])
    assert abs(loss.item() - ref_loss) <= 2e-3 * ref_loss
    grads = {"latent": lat.grad}
    grads.update({"coarse." + k: v.grad for k, v in net.mlp_coarse.named_parameters()})
    grads.update({"fine." + k: v.grad for k, v in net.mlp_fine.named_parameters()})
    assert len(grads) == 61
    for key, gr in grads.items():
        flat = gr.detach().reshape(-1).cpu().numpy()
        ref_s, ref_n = gg[f"{name}_grad_{key}_sample"], float(gg[f"{name}_grad_{key}_norm"])
        got_s = flat[synthetic.grad_sample_index(flat.size, key)]
        assert abs(np.linalg.norm(flat.astype(np.float64)) - ref_n) <= 3e-2 * ref_n, key
        assert np.linalg.norm(got_s - ref_s) <= 3e-2 * np.linalg.norm(ref_s), f"{key}: {np.linalg.norm(got_s - ref_s) / np.linalg.norm(ref_s):.3e}"


@pytest.mark.parametrize("scene_name,B", [("train", 96), ("mv_mini", 96), ("train", 77), ("mv_mini", 45)])  # 77 / 45: ragged last tile
def test_direct_forward_is_differentiable(dev, scene_name, B):
    """net(xyz, viewdirs) with grad enabled (src/model/models.py:146-266 under autograd): parameter and latent-grid
    gradients of a random linear functional of the (rgb, sigma) outputs, against torch autograd through the oracle."""
    from helpers import mlp_params, scene_for
    from test_api_gpu import build_net
    scene, meta = scene_for(scene_name)
    SB = scene["SB"]
    gen = torch.Generator().manual_seed(13)
    xyz = (torch.rand(SB, B, 3, generator=gen) - 0.5) * 1.6
    vd = torch.nn.functional.normalize(torch.randn(SB, B, 3, generator=gen), dim=-1)
    gw = torch.randn(SB, B, 4, generator=gen)
    p = {k: v.clone().requires_grad_(True) for k, v in mlp_params(11).items()}
    sc = dict(scene)
    sc["latent"] = scene["latent"].clone().requires_grad_(True)
    ref = O.pixelnerf_forward(sc, p, xyz, vd)
    (ref * gw).sum().backward()
    net = build_net(dev, scene, precision="f16").train()
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    out = net(xyz.to(dev), coarse=True, viewdirs=vd.to(dev))
    assert out.shape == (SB, B, 4) and out.requires_grad
    assert (out.detach().cpu() - ref.detach()).abs().max() <= 2e-2
    (out * gw.to(dev)).sum().backward()
    pairs = [("latent", lat.grad.cpu(), sc["latent"].grad)]
    pairs += [(k, v.grad.cpu(), p[k].grad) for k, v in net.mlp_coarse.named_parameters()]
    assert all(v.grad is None for v in net.mlp_fine.parameters())
    for k, a, b in pairs:
        rel = float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
        cos = float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm() + 1e-30))
        # 16-bit operand chain on 96 points per object (no averaging over a batch of rays): 5e-2 / 0.998
        assert rel <= 5e-2 and cos >= 0.998, f"{k}: rel err {rel:.3e}, cos {cos:.6f}"
    with pytest.raises(NotImplementedError):
        net(xyz.to(dev).requires_grad_(True), coarse=True, viewdirs=vd.to(dev))


def test_fused_optimizer_updates_reach_the_kernels(dev):
    """torch.optim.Adam(fused=True) writes the parameters in place without bumping tensor._version: the packed streams
    must be rebuilt anyway.  Same batch, same frozen noise, six steps with the fused and with the foreach form of Adam:
    both must lower the loss, along the same trajectory."""
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util import DotMap
    from pixelnerf_amd.util.conf import default_model_conf
    from helpers import mlp_params
    g, scene, meta, mc, mf, rays, noise = golden_setup("train_64_32")
    gt = torch.rand(4, 32, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) * 0.5 + 0.25
    r = rays.to(dev)
    traj = {}
    for fused in (False, True):
        net = make_model(default_model_conf(), precision="f16").to(dev).train()
        net.mlp_coarse.load_state_dict(mlp_params(11))
        net.mlp_fine.load_state_dict(mlp_params(12))
        net.encoder.latent = scene["latent"].to(dev)
        ls = torch.tensor([32.0, 32.0], device=dev)
        net.encoder.latent_scaling = ls / (ls - 1) * 2.0
        net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
        net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
        net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
        rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).to(dev)
        render_par = rend.bind_parallel(net, None, simple_output=False).train()
        opt = torch.optim.Adam(list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters()), lr=2e-4, fused=fused)
        losses = []
        for it in range(6):
            torch.manual_seed(123)
            rd = DotMap(render_par(r, want_weights=True))
            loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        traj[fused] = losses
        assert losses[-1] < 0.9 * losses[0], (fused, losses)
    for a, b in zip(traj[False], traj[True]):
        assert abs(a - b) <= 2e-2 * abs(a), traj


@pytest.mark.parametrize("scale", [1e-20, 1.0, 1e20])
def test_latent_scatter_follows_the_gradient_scale(dev, scale):
    """The LDS slab of the latent scatter (fp64 since round 6; 64-bit fixed point at 2^40 / max |gradient| before) has no
    scale of its own: the result must be the same relative to the input scale from 1e-20 to 1e20, zeros stay zeros, and
    a few huge entries must not wipe out ordinary ones."""
    from helpers import scene_for
    from pixelnerf_amd import ops
    from testdata import synthetic
    scene, meta = scene_for("mv_mini")
    sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev),
                        scene["image_shape"], scene["NS"])
    gen = torch.Generator().manual_seed(5)
    r = synthetic.target_rays(meta, n_rays=40).reshape(-1, 8)
    z = O.sample_coarse(r, torch.rand(r.shape[0], 12, generator=gen), 12)
    rows = scene["NS"] * r.shape[0] * 12
    d = torch.randn(rows, 512, generator=gen)
    d[::7] *= 1e4   # a wide dynamic range inside one slab
    d[3] = 0.0
    Hl, Wl = scene["latent"].shape[-2:]
    ref = ops.latent_scatter(sc, r.to(dev), z.to(dev), d.to(dev), torch.zeros(4, Hl, Wl, 512, device=dev))
    got = ops.latent_scatter(sc, r.to(dev), z.to(dev), (d * scale).to(dev), torch.zeros(4, Hl, Wl, 512, device=dev))
    assert torch.isfinite(got).all()
    err = (got / scale - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item(), err
    zero = ops.latent_scatter(sc, r.to(dev), z.to(dev), torch.zeros_like(d).to(dev), torch.zeros(4, Hl, Wl, 512, device=dev))
    assert not zero.any()
