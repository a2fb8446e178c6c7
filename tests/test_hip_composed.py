"""
GPU tests (-m gpu): everything the reference's model classes accept OUTSIDE the one configuration the fused kernels implement.

  * `pnr_linear` / `pnr_linear_backward` (include/pixelnerf_hip.h): one nn.Linear with the ReLU in front and the residual behind,
    any rows / d_in / d_out, against torch fp32 on the CPU -- exact-fp32 form and the fp32-class split-operand form;
  * `ResnetFC.forward` of arbitrary constructor arguments (src/model/resnetfc.py:66-184: widths, block counts, combine layers,
    Softplus, SPADE, d_in = 0, d_latent = 0, view mean / maximum), outputs AND gradients against the oracle's general restatement
    under torch autograd;
  * `PixelNeRFNet.forward` under the model confs of testdata.synthetic.VARIANTS (models.py:22-65: coded view directions -- the
    reference's default --, camera-space positions, depth-only feature, global encoder, no encoder ...) against
    tests/golden/variants.npz, which oracle/make_goldens.py froze from the UNMODIFIED reference; gradients (parameters and the
    latent grid) against the oracle's autograd;
  * `NeRFRenderer` around such a network: it takes the reference's control flow around the model callable -- rendered and trained.

Tolerances.  Exact form: 2e-6 relative to the tensor's scale (fp32 sums in another order).  fp32-class form (the default): 2e-5
(tail x tail products dropped: 2^-22 per product, like the fused fp32-class kernels); gradients 1e-4 relative per tensor.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import pnr_oracle as O
from testdata import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def maxrel(a, b):
    a, b = a.detach().cpu().double().reshape(-1), b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


# ------------------------------------------------------------------------------------------------ one nn.Linear
@pytest.mark.parametrize("precision,tol", [("f32", 2e-6), ("f16x3", 2e-5)])
@pytest.mark.parametrize("rows,d_in,d_out,relu_in,residual,bias", [
    (1, 3, 4, False, False, True),
    (200, 78, 512, False, False, True),
    (4099, 96, 96, True, True, True),       # no multiple of any tile size in any dimension
    (777, 512, 4, True, False, True),       # lin_out's shape
    (513, 528, 128, False, True, False),    # 512 + 16 latent columns, no bias
    (130, 1, 64, False, False, True),       # K = 1
])
def test_linear_operator_and_its_backward_match_torch(dev, precision, tol, rows, d_in, d_out, relu_in, residual, bias):
    from pixelnerf_amd import ops
    g = torch.Generator().manual_seed(rows * 7 + d_in)
    x = torch.randn(rows, d_in, generator=g)
    w = torch.randn(d_out, d_in, generator=g) * (2.0 / d_in) ** 0.5
    b = torch.randn(d_out, generator=g) * 0.1 if bias else None
    r = torch.randn(rows, d_out, generator=g) if residual else None
    dy = torch.randn(rows, d_out, generator=g) * 1e-3    # training-sized gradients: exercises the power-of-two gradient scale
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    br = None if b is None else b.clone().requires_grad_(True)
    ref = torch.nn.functional.linear(torch.relu(xr) if relu_in else xr, wr, br)
    if r is not None:
        ref = ref + r
    ref.backward(dy)

    y = ops.linear(x.to(dev), w.to(dev), None if b is None else b.to(dev), relu_in=relu_in, residual=None if r is None else r.to(dev),
                   precision=precision)
    assert y.shape == (rows, d_out)
    assert maxrel(y, ref) <= tol, maxrel(y, ref)
    dx, dw, db = ops.linear_backward(dy.to(dev), x.to(dev), w.to(dev), relu_in=relu_in, need_db=bias, precision=precision)
    assert maxrel(dx, xr.grad) <= tol * 5, ("dx", maxrel(dx, xr.grad))
    assert maxrel(dw, wr.grad) <= tol * 5, ("dw", maxrel(dw, wr.grad))
    if bias:
        assert maxrel(db, br.grad) <= tol * 5, ("db", maxrel(db, br.grad))
    # partial requests
    dx2, dw2, db2 = ops.linear_backward(dy.to(dev), x.to(dev), w.to(dev), relu_in=relu_in, need_dx=False, need_db=False, precision=precision)
    assert dx2 is None and db2 is None and torch.equal(dw2, dw)
    dx3, dw3, _ = ops.linear_backward(dy.to(dev), x.to(dev), w.to(dev), relu_in=relu_in, need_dw=False, need_db=False, precision=precision)
    assert dw3 is None and torch.equal(dx3, dx)


def test_linear_operator_is_bit_reproducible_and_loud(dev):
    from pixelnerf_amd import _lib, ops
    g = torch.Generator().manual_seed(3)
    x, w, dy = (torch.randn(3000, 96, generator=g).to(dev), torch.randn(40, 96, generator=g).to(dev), torch.randn(3000, 40, generator=g).to(dev))
    a = ops.linear_backward(dy, x, w, relu_in=True)
    b = ops.linear_backward(dy, x, w, relu_in=True)
    assert all(torch.equal(p, q) for p, q in zip(a, b))  # fixed-order slice reduction: no atomics
    with pytest.raises(_lib.PixelNerfHipError):
        ops.linear(x.cpu(), w, None)
    with pytest.raises(ValueError):
        ops.linear(x, w, None, precision="f16")
    with pytest.raises(ValueError):
        ops.linear(x[:, :50], w, None)
    assert ops.linear(x[:0], w, None).shape == (0, 40)


# ------------------------------------------------------------------------------------------------ ResnetFC of any shape
RESNETFC_CASES = {
    # name: (d_in, d_latent, kwargs, rows layout (groups, NS, B))
    "narrow_mean":   (16, 32, dict(d_hidden=64, n_blocks=3, combine_layer=1, combine_type="average"), (2, 3, 17)),
    "softplus_spade_max": (42, 48, dict(d_hidden=96, n_blocks=4, combine_layer=2, combine_type="max", beta=3.0, use_spade=True), (1, 2, 33)),
    "no_input":      (0, 24, dict(d_hidden=32, n_blocks=2, combine_layer=2), (1, 1, 50)),          # d_in = 0: x starts as zeros
    "no_latent":     (5, 0, dict(d_hidden=128, n_blocks=2, combine_layer=1), (2, 2, 20)),          # d_latent = 0: no lin_z, still pooled
    "never_pooled":  (7, 9, dict(d_hidden=40, n_blocks=2), (1, 2, 11)),                            # combine_layer 1000 > n_blocks
    "shipped_shape": (42, 512, dict(d_hidden=512, n_blocks=5, combine_layer=3), (1, 2, 40)),       # differentiable twin of the fused shape
}


@pytest.mark.parametrize("case", sorted(RESNETFC_CASES))
@pytest.mark.parametrize("precision", ["f16x3", "f32"])
def test_resnetfc_of_any_shape_matches_oracle_with_gradients(dev, case, precision):
    from pixelnerf_amd.model.resnetfc import ResnetFC
    d_in, d_latent, kw, (G, NS, B) = RESNETFC_CASES[case]
    if case == "shipped_shape" and precision == "f32":
        pytest.skip("one precision is enough for the largest case")
    mlp = ResnetFC(d_in, d_latent=d_latent, **kw)
    shapes = [(k, tuple(v.shape)) for k, v in mlp.state_dict().items()]
    assert shapes == synthetic.resnetfc_shapes(d_in, d_latent, **kw)  # the reference's state_dict layout
    params = synthetic.fill_state(shapes, 5)
    mlp.load_state_dict(params)
    mlp = mlp.to(dev)
    mlp.composed_precision = precision
    dims = (NS, B)
    pooled = kw.get("combine_layer", 1000) < kw["n_blocks"]
    # A ReLU input within rounding of zero may take the other branch in another arithmetic class (the fp32-class operators agree with
    # fp32 to ~1e-6, not bit for bit), which moves every upstream gradient by O(1 / sqrt(elements)) -- measured here once: ONE unit of
    # 20 480 flipped, 1.4e-3 on all gradients while every single operator was exact to 7e-7 on the same inputs
    # (a one-off probe of round 4, tools/gpu_debug_composed3.py in the git history).  The comparison is about the operators: of 19 seeded draws the input that stays FARTHEST from
    # a kink (min |relu input| / rms of its layer, from the oracle) is used -- ~1.6e-5 for the 450 000 ReLU inputs of the largest case.
    best = None
    for seed in range(1, 20):
        zx_try = torch.randn(G * NS * B, d_latent + d_in, generator=torch.Generator().manual_seed(seed)) * 0.7
        margin = []
        with torch.no_grad():
            O.resnetfc_forward_general(params, zx_try, dims, d_in, d_latent, kw["d_hidden"], kw["n_blocks"], kw.get("combine_layer", 1000),
                                       kw.get("combine_type", "average"), kw.get("beta", 0.0), kw.get("use_spade", False), kink_margin=margin)
        m = min(margin) if margin else float("inf")
        if best is None or m > best[0]:
            best = (m, seed, zx_try)
        if m > 1e-4:
            break
    g = torch.Generator().manual_seed(100 + best[1])
    zx = best[2]
    cpu = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    zx_ref = zx.clone().requires_grad_(True)
    ref = O.resnetfc_forward_general(cpu, zx_ref, dims, d_in, d_latent, kw["d_hidden"], kw["n_blocks"], kw.get("combine_layer", 1000),
                                     kw.get("combine_type", "average"), kw.get("beta", 0.0), kw.get("use_spade", False))
    w_out = torch.randn(ref.shape, generator=g)
    (ref * w_out).sum().backward()

    zx_dev = zx.to(dev).requires_grad_(True)
    out = mlp(zx_dev, combine_inner_dims=dims)
    assert out.shape == ref.shape and out.numel() == 4 * (G * B if pooled else G * NS * B)  # pooled: (groups, B, 4), util.py:464-466
    tol = 2e-6 if precision == "f32" else 2e-5
    assert maxrel(out, ref) <= tol, maxrel(out, ref)
    (out * w_out.to(dev)).sum().backward()
    gtol = 2e-5 if precision == "f32" else 1e-4
    worst = rel(zx_dev.grad, zx_ref.grad)
    assert worst <= gtol, ("zx", worst)
    for k, p in mlp.named_parameters():
        e = rel(p.grad, cpu[k].grad)
        worst = max(worst, e)
        assert e <= gtol, (k, e)
    print(f"ResnetFC {case} ({precision}): out {maxrel(out, ref):.1e}, worst relative gradient error {worst:.1e}")
    with torch.no_grad():
        again = mlp(zx.to(dev), combine_inner_dims=dims)
    assert not again.requires_grad and maxrel(again, ref) <= max(tol, 2e-5)


def test_resnet_block_with_shortcut(dev):
    """size_in != size_out (resnetfc.py:47-50,59-62: x_s = shortcut(x)); the reference's own constructor trips over the bias of its
    bias-free shortcut, so the pin is the formula itself in torch"""
    from pixelnerf_amd.model.resnetfc import ResnetBlockFC
    blk = ResnetBlockFC(48, 80, 32)
    with torch.no_grad():
        blk.fc_1.weight.normal_(0, 0.05)
    x = torch.randn(301, 48, generator=torch.Generator().manual_seed(2))
    F = torch.nn.functional
    net = F.linear(torch.relu(x), blk.fc_0.weight, blk.fc_0.bias)
    ref = F.linear(x, blk.shortcut.weight) + F.linear(torch.relu(net), blk.fc_1.weight, blk.fc_1.bias)
    out = blk.to(dev)(x.to(dev))
    assert maxrel(out, ref) <= 2e-5


# ------------------------------------------------------------------------------------------------ PixelNeRFNet variants
def build_variant(dev, name, precision="f16x3"):
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.util.conf import Conf
    g = load_golden("variants")
    scene, meta, xyz, vd, glob = synthetic.variant_inputs(name)
    net = make_model(Conf(synthetic.variant_model_conf(name)), precision=precision).eval()
    assert net.d_in == int(g[f"{name}_d_in"]) and net.d_latent == int(g[f"{name}_d_latent"])
    pc, pf = synthetic.variant_mlp_params(name, net.d_in, net.d_latent)
    net.mlp_coarse.load_state_dict(pc)
    net.mlp_fine.load_state_dict(pf)
    net = net.to(dev)
    lat = scene["latent"].to(dev)
    net.encoder.latent = lat
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32, device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    if glob is not None:
        net.global_encoder.latent = glob.to(dev)
    return net, scene, meta, xyz, vd, glob, (pc, pf), g


@pytest.mark.parametrize("name", sorted(synthetic.VARIANTS))
def test_model_variants_match_reference_outputs(dev, name):
    net, scene, meta, xyz, vd, glob, _, g = build_variant(dev, name)
    assert not net.fused_supported()
    with torch.no_grad():
        for which, coarse in (("coarse", True), ("fine", False)):
            out = net(xyz.to(dev), coarse=coarse, viewdirs=vd.to(dev)).cpu().numpy()
            ref = g[f"{name}_out_{which}"]
            assert out.shape == ref.shape
            np.testing.assert_allclose(out[..., :3], ref[..., :3], rtol=0, atol=3e-5)
            np.testing.assert_allclose(out[..., 3], ref[..., 3], rtol=2e-4, atol=3e-4)
    with pytest.raises(NotImplementedError):
        net._check_supported()  # the fused network's entry points stay closed to it


@pytest.mark.parametrize("name", ["code_viewdirs", "softplus_spade_max", "global_encoder", "no_normalize_z"])
def test_model_variants_are_differentiable(dev, name):
    """parameters and the latent grid: the composed forward's HIP backward operators (pnr_linear_backward, pnr_grid_index_backward,
    pnr_positional_encoding_backward) against torch autograd through the oracle"""
    net, scene, meta, xyz, vd, glob, (pc, pf), _ = build_variant(dev, name)
    conf = synthetic.variant_model_conf(name)
    net.train()
    lat_dev = net.encoder.latent.detach().clone().requires_grad_(True)
    net.encoder.latent = lat_dev
    out = net(xyz.to(dev), coarse=True, viewdirs=vd.to(dev))
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    (out * w.to(dev)).sum().backward()

    cpu = {k: v.clone().requires_grad_(True) for k, v in pc.items()}
    sc = dict(scene)
    sc["latent"] = scene["latent"].clone().requires_grad_(True)
    ref = O.pixelnerf_forward_general(sc, cpu, xyz, vd, conf, global_latent=glob)
    (ref * w).sum().backward()
    assert maxrel(out, ref) <= 3e-5
    worst = rel(lat_dev.grad, sc["latent"].grad)
    assert worst <= 1e-4, ("latent", worst)
    for k, p in net.mlp_coarse.named_parameters():
        e = rel(p.grad, cpu[k].grad)
        worst = max(worst, e)
        assert e <= 1e-4, (k, e)
    assert all(p.grad is None for p in net.mlp_fine.parameters())
    print(f"variant {name}: worst relative gradient error {worst:.1e}")


def test_renderer_around_a_model_variant_renders_and_trains(dev):
    """NeRFRenderer with a PixelNeRFNet the fused kernels do not cover: the reference's control flow around net(xyz, coarse=,
    viewdirs=) with the HIP renderer kernels; against the same control flow on the CPU (oracle stage functions around the oracle's
    general forward), forward and parameter gradients"""
    from pixelnerf_amd.render import NeRFRenderer
    from test_hip_generic_training import loss_of, oracle_render
    name = "code_viewdirs"
    net, scene, meta, xyz, vd, glob, (pc, pf), _ = build_variant(dev, name)
    conf = synthetic.variant_model_conf(name)
    SB, B, Kc, Kf, Kfd = scene["SB"], 24, 16, 16, 4
    rays = synthetic.target_rays(meta, n_rays=B)
    R = SB * B
    noise = synthetic.make_noise(R, Kc, Kf, Kfd)
    g = torch.Generator().manual_seed(9)
    tgt = {"rgb": torch.rand(R, 3, generator=g), "depth": torch.rand(R, generator=g) * 2 + 1,
           "w": [torch.randn(R, Kc + Kf, generator=g), torch.randn(R, Kc + Kf, generator=g)]}
    renderer = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=0.05, white_bkgd=True, eval_batch_size=500).to(dev).train()
    net.train()
    out = renderer(net, rays.to(dev), want_weights=True, _noise={k: v.to(dev) for k, v in noise.items()})
    got = {p: dict(rgb=out[p].rgb, depth=out[p].depth, weights=out[p].weights) for p in ("coarse", "fine")}
    loss = loss_of(got, {"rgb": tgt["rgb"].to(dev), "depth": tgt["depth"].to(dev), "w": [t.to(dev) for t in tgt["w"]]})
    loss.backward()

    cpu = [{k: v.clone().requires_grad_(True) for k, v in p.items()} for p in (pc, pf)]

    class Ref(torch.nn.Module):
        use_viewdirs = True

        def forward(self, pts, coarse=True, viewdirs=None):
            return O.pixelnerf_forward_general(scene, cpu[0] if coarse else cpu[1], pts, viewdirs, conf)
    ref = oracle_render(Ref(), rays, noise, Kc, Kf, Kfd, 0.05, True, False,
                        wc_for_sampling=got["coarse"]["weights"].detach().cpu().reshape(R, Kc))
    ref_loss = loss_of(ref, tgt)
    ref_loss.backward()
    for p in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            a, b = got[p][k].detach().cpu().reshape(-1), ref[p][k].detach().reshape(-1)
            assert (a - b).abs().max() <= 5e-5 * max(1.0, float(b.abs().max())), (p, k, float((a - b).abs().max()))
    worst = 0.0
    for mlp, ref_p in ((net.mlp_coarse, cpu[0]), (net.mlp_fine, cpu[1])):
        for k, p in mlp.named_parameters():
            e = rel(p.grad, ref_p[k].grad)
            worst = max(worst, e)
            assert e <= 2e-3, (k, e)
    print(f"renderer around model variant {name}: worst relative gradient error {worst:.1e}")


def test_global_encoder_conf_end_to_end_through_encode(dev):
    """use_global_encoder=True through the public flow -- encode(images, poses, focal) runs BOTH trunks (models.py:143-144), the
    global latent rides in front of the pixel-aligned one (models.py:228-235), and a loss on the output reaches the ResnetFC, the
    global encoder's projection AND both ResNet trunks (stop_encoder_grad=False: train/train.py trains the encoder too)"""
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.util.conf import Conf
    conf = Conf(synthetic.variant_model_conf("global_encoder"))
    torch.manual_seed(0)
    net = make_model(conf).to(dev).train()
    with torch.no_grad():
        for m in (net.mlp_coarse, net.mlp_fine):
            for b in m.blocks:
                b.fc_1.weight.normal_(0, 0.03)   # the reference zero-initialises fc_1: perturbed so that every term is live
    g = torch.Generator().manual_seed(3)
    images = (torch.rand(1, 2, 3, 64, 64, generator=g) * 2 - 1).to(dev)
    poses = torch.stack([synthetic.pose_spherical(30.0, -20.0, 2.7), synthetic.pose_spherical(80.0, -10.0, 2.7)])[None].to(dev)
    net.encode(images, poses, torch.tensor(119.4, device=dev))
    assert net.num_views_per_obj == 2 and tuple(net.global_encoder.latent.shape) == (2, 16) and net.encoder.latent.requires_grad
    xyz = (torch.rand(1, 50, 3, generator=g) - 0.5).to(dev)
    vd = torch.nn.functional.normalize(torch.randn(1, 50, 3, generator=g), dim=-1).to(dev)
    out = net(xyz, coarse=True, viewdirs=vd)
    assert out.shape == (1, 50, 4) and bool(torch.isfinite(out).all())
    out.square().mean().backward()
    for name in ("mlp_coarse.lin_z.0.weight", "mlp_coarse.lin_in.weight", "global_encoder.fc.weight", "global_encoder.model.conv1.weight",
                 "encoder.model.conv1.weight"):
        p = dict(net.named_parameters())[name]
        assert p.grad is not None and float(p.grad.abs().max()) > 0, name
    assert net.mlp_coarse.lin_z[0].weight.shape == (128, 512 + 16)
    # the same latent columns, assembled by hand, through the ResnetFC alone give the same output
    with torch.no_grad():
        again = net(xyz, coarse=True, viewdirs=vd)
    assert (again - out.detach()).abs().max() <= 1e-6


@pytest.mark.parametrize("interp,padding", [("bilinear", "zeros"), ("nearest", "border"), ("bilinear", "reflection")])
def test_encoder_lookup_modes_other_than_the_shipped_one_run_composed(dev, interp, padding):
    """the fused kernels hard-code grid_sample(bilinear, border); a conf with any other `index_interp` / `index_padding`
    the reference honours (encoder.py:27-28,100-108) must NOT run them.  It goes down the composed forward, whose lookup is then
    ATen's grid_sample on the HIP tensors -- outputs and the grid gradient against the oracle with the same modes."""
    from helpers import mlp_params
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.util.conf import Conf
    conf = synthetic.variant_model_conf("code_viewdirs")
    shipped = dict(type="resnet", n_blocks=5, d_hidden=512, combine_layer=3, combine_type="average")
    conf.update(use_code_viewdirs=False, mlp_coarse=dict(shipped), mlp_fine=dict(shipped),
                encoder=dict(conf["encoder"], index_interp=interp, index_padding=padding))
    net = make_model(Conf(conf)).eval()
    assert net.encoder.index_interp == interp and not net.fused_supported()
    pc = mlp_params(11)
    net.mlp_coarse.load_state_dict(pc)
    net = net.to(dev)
    scene, meta, xyz, vd, _ = synthetic.variant_inputs("no_normalize_z")  # seeded points, four of them far off the source image
    lat = scene["latent"].to(dev).requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32, device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    out = net(xyz.to(dev), coarse=True, viewdirs=vd.to(dev))
    sc = dict(scene)
    sc["latent"] = scene["latent"].clone().requires_grad_(True)
    ref = O.pixelnerf_forward_general(sc, pc, xyz, vd, conf)
    border = O.pixelnerf_forward_general(sc, pc, xyz, vd, dict(conf, encoder={}))
    assert (ref - border).abs().max() > 1e-3, "the fixture must tell the modes apart"
    assert maxrel(out, ref) <= 3e-5
    w = torch.randn(out.shape, generator=torch.Generator().manual_seed(4))
    (out * w.to(dev)).sum().backward()
    (ref * w).sum().backward()
    assert rel(lat.grad, sc["latent"].grad) <= 1e-4
