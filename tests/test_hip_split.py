"""
GPU parity tests (-m gpu) of the fp32-CLASS fast path (precision "f16x3", pixel-nerf_amd/csrc/pnr_split.hip): the fused
network kernel with every operand carried as an fp16 (head, tail) pair -- three f16 MFMAs per product, fp32 accumulate,
lin_z folded into fp32 tables.  It is held to EXACTLY the bars of the exact-fp32 validation path (tests/test_hip_f32.py)
against the reference's own fp32 outputs (tests/golden, identical rays, weights, grid, noise):
  * per point : |rgb| err <= 2e-5, sigma err <= 1e-4 * max(1, sigma)
  * renders   : coarse rgb <= 2e-5, depth <= 1e-4 (far-near), weights <= 2e-5; the fine pass within the same bounds except
                for a <= 2 % allowance of rays whose importance samples flipped a cdf bin at rounding level; PSNR >= 85 dB.
Single- and multi-view scenes run 64-point tiles (multi-view: the running view sum is parked in an L2-resident scratch).
"""
import numpy as np
import pytest
import torch

from helpers import RENDER_SCENARIOS, assert_close_frac, golden_setup, load_golden, mlp_params, scene_for, MV_STAGE_SCENES, STAGE_SCENES
from oracle import pnr_oracle as O

pytestmark = pytest.mark.gpu

SCENARIOS = list(RENDER_SCENARIOS)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from pixelnerf_amd import ops as _ops
    return _ops


def dscene(ops, dev, name):
    s, _ = scene_for(name)
    return ops.make_scene(s["latent"].to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])


def split_net(ops, dev, sc, seed):
    state = {k: v.to(dev) for k, v in mlp_params(seed).items()}
    return ops.pack_mlp(state, "f16x3"), ops.fold_latent(sc, state, "f16x3")


def test_split_tables_are_lin_z_of_the_grid_in_fp32(ops, dev):
    s, _ = scene_for("sn64")
    sc = dscene(ops, dev, "sn64")
    p = mlp_params(12)
    tab = ops.fold_latent(sc, {k: v.to(dev) for k, v in p.items()}, "f16x3")
    assert tab.dtype == torch.float32
    perm = ops.storage_perm().long()
    grid = s["latent"].permute(0, 2, 3, 1).double()
    for b in range(3):
        ref = grid @ p[f"lin_z.{b}.weight"].double().t() + p[f"lin_z.{b}.bias"].double()
        got = torch.empty_like(ref)
        got[..., perm] = tab[b].cpu().double()
        assert (got - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("shape", [(3, 75, 101), (1, 128, 128), (2, 64, 64), (1, 90, 91), (1, 64, 100)])
def test_split_tables_of_large_grids(ops, dev, shape):
    """grids of 8192 texels and more take fold_split_big_kernel (256 x 256 tiles, double-buffered swizzled LDS images): the same bar
    as the small-grid kernel against an fp64 product, on ragged texel counts (22 725 = 88 x 256 + 197; 8190: the 128 x 128 kernel
    just below the crossover), exact multiples of the tile (16 384, 8192) and a count whose last row tile holds one texel row group
    (6400 = 25 x 256)"""
    from pixelnerf_amd import _lib
    n, Hl, Wl = shape
    gen = torch.Generator().manual_seed(4)
    lat = torch.randn(n, 512, Hl, Wl, generator=gen)
    lat[0, :, 0, 0] *= 50.0  # a texel with large entries: the (head, tail) split carries them
    s, _ = scene_for("dtu_mini")
    sc = ops.make_scene(lat.to(dev), s["poses"][:n].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], n)
    p = mlp_params(12)
    tab = ops.fold_latent(sc, {k: v.to(dev) for k, v in p.items()}, "f16x3")
    perm = ops.storage_perm().long()
    grid = lat.permute(0, 2, 3, 1).double()
    for b in range(3):
        ref = grid @ p[f"lin_z.{b}.weight"].double().t() + p[f"lin_z.{b}.bias"].double()
        got = torch.empty_like(ref)
        got[..., perm] = tab[b].cpu().double()
        assert (got - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max())), (shape, b, float((got - ref).abs().max()))


@pytest.mark.parametrize("scene_name", STAGE_SCENES)
def test_split_eval_points_matches_reference(ops, dev, scene_name):
    g = load_golden("stages")
    sc = dscene(ops, dev, scene_name)
    xyz = torch.from_numpy(g[f"{scene_name}_xyz"]).to(dev)
    vd = torch.from_numpy(g[f"{scene_name}_viewdirs"]).to(dev)
    for which, seed in (("coarse", 11), ("fine", 12)):
        pk, tab = split_net(ops, dev, sc, seed)
        out = ops.eval_points(sc, pk, xyz, vd, tables=tab).cpu().numpy()
        ref = g[f"{scene_name}_out_{which}"]
        assert np.isfinite(out).all()
        e_rgb = np.abs(out[..., :3] - ref[..., :3]).max()
        e_s = (np.abs(out[..., 3] - ref[..., 3]) / np.maximum(1.0, ref[..., 3])).max()
        print(f"SPLIT eval_points {scene_name} {which}: rgb max err {e_rgb:.3e}, sigma rel err {e_s:.3e}")
        assert e_rgb <= 2e-5, f"rgb max err {e_rgb:.3e}"
        assert e_s <= 1e-4, f"sigma rel err {e_s:.3e}"


@pytest.mark.parametrize("name", SCENARIOS)
def test_split_render_matches_reference(ops, dev, name):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    sc = dscene(ops, dev, str(g["scene"]))
    pc, tc = split_net(ops, dev, sc, int(g["mlp_seed_coarse"]))
    pf, tf = split_net(ops, dev, sc, int(g["mlp_seed_fine"])) if mf is not None else (None, None)
    r = rays.reshape(-1, 8).to(dev)
    out = ops.render_forward(sc, pc, pf, r, Kc, Kf, Kfd, {k: v.to(dev) for k, v in noise.items()},
                             depth_std=float(g["depth_std"]), white_bkgd=bool(g["white_bkgd"]),
                             lindisp=bool(g["lindisp"]), want_weights=True, tables=(tc, tf))
    span = float(meta["z_far"] - meta["z_near"])
    assert ("fine" in out) == (Kf > 0)
    for p in ["coarse"] + (["fine"] if Kf > 0 else []):
        K = Kc if p == "coarse" else Kc + Kf
        flips = 0.0 if p == "coarse" else 2e-2
        rgb, depth, w = out[p]["rgb"].cpu(), out[p]["depth"].cpu().numpy(), out[p]["weights"].cpu().numpy()
        assert_close_frac(rgb.numpy(), g[f"{p}_rgb"].reshape(-1, 3), 2e-5, max_frac=flips, loose_atol=0.05, what=f"{p} rgb")
        assert_close_frac(depth, g[f"{p}_depth"].reshape(-1), 1e-4 * span, max_frac=flips, loose_atol=0.05 * span, what=f"{p} depth")
        if p == "coarse":
            np.testing.assert_allclose(w, g["coarse_weights"].reshape(-1, K), rtol=0, atol=2e-5)
        ps = O.psnr(rgb, torch.from_numpy(g[f"{p}_rgb"]).reshape(-1, 3))
        print(f"SPLIT render {name} {p}: PSNR {ps:.1f} dB")
        assert ps >= 85.0, f"{p} PSNR {ps:.1f} dB"


def test_split_variants_agree_and_match_the_exact_fp32_path_at_full_size(ops, dev):
    """65 536 samples-rays of the sn64 view: ray-sample and explicit-point variants agree bitwise; against the unfused
    fp32-MFMA path (a different implementation of the same arithmetic class) the per-point error stays at 1e-5."""
    from testdata import synthetic
    s, meta = scene_for("sn64")
    sc = dscene(ops, dev, "sn64")
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    pk, tab = ops.pack_mlp(state, "f16x3"), ops.fold_latent(sc, state, "f16x3")
    rays = synthetic.target_rays(meta).reshape(-1, 8).to(dev)
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], 64, device=dev))
    a = ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
    assert torch.equal(a, ops.eval_ray_samples(sc, pk, rays, z, tables=tab))
    pts = (rays[:, None, :3] + z.unsqueeze(2) * rays[:, None, 3:6]).reshape(1, -1, 3)
    vd = rays[:, None, 3:6].expand(-1, 64, -1).reshape(1, -1, 3)
    b = ops.eval_points(sc, pk, pts.contiguous(), vd.contiguous(), tables=tab).reshape(a.shape)
    assert torch.equal(a, b)
    exact = ops.eval_ray_samples(sc, ops.pack_mlp(state, "f32"), rays, z)
    e = (a - exact).abs()
    print(f"SPLIT vs exact-fp32 path, {a.shape[0] * a.shape[1]} points: rgb max {e[..., :3].max().item():.3e}, "
          f"sigma max {e[..., 3].max().item():.3e}")
    assert e[..., :3].max().item() <= 2e-5 and (e[..., 3] / exact[..., 3].clamp(min=1.0)).max().item() <= 1e-4


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
@pytest.mark.parametrize("name,nrays", [("sn64", 4096), ("sn64", 120), ("sn64", 37), ("mv_mini", 776)])
def test_tile_order_does_not_touch_a_bit(ops, dev, monkeypatch, precision, name, nrays):
    """pnr_device.h tile_range(): the XCD-aware order (device XCD count), the plain grid-stride order (PIXELNERF_XCD_COUNT=0)
    and XCD counts that do / do not divide the grid (4, 3, 32) visit every tile exactly once: identical outputs, no tile left
    at its initial value -- on grids that fill the chip, on short ones (fewer tiles than XCDs x CUs) and on multi-view scenes."""
    from testdata import synthetic
    s, meta = scene_for(name)
    sc = dscene(ops, dev, name)
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    pk, tab = ops.pack_mlp(state, precision, folded=precision == "f16"), ops.fold_latent(sc, state, precision)
    rays = synthetic.target_rays(meta).reshape(-1, 8)[:nrays].contiguous().to(dev)
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], 64, device=dev))
    monkeypatch.delenv("PIXELNERF_XCD_COUNT", raising=False)
    ref = ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
    assert torch.isfinite(ref).all()
    for n in ("0", "1", "3", "4", "8", "32"):
        monkeypatch.setenv("PIXELNERF_XCD_COUNT", n)
        assert torch.equal(ref, ops.eval_ray_samples(sc, pk, rays, z, tables=tab)), f"tile order with {n} XCDs"


def test_split_api_scope(ops, dev):
    """PixelNeRFNet(precision='f16x3'): single- and multi-view scenes run the split kernel; a 16-bit table set is refused."""
    from pixelnerf_amd import _lib
    from test_api_gpu import build_net
    from pixelnerf_amd.render import NeRFRenderer
    g, scene, meta, mc, mf, rays, noise = golden_setup("sn64_64_128")
    net = build_net(dev, scene, precision="f16x3")
    assert net.packed(True).precision == _lib.PREC_F16X3 and net.tables(True).dtype == torch.float32
    rend = NeRFRenderer(n_coarse=64, n_fine=128, n_fine_depth=16, white_bkgd=True).to(dev).eval()
    with torch.no_grad():
        out = rend(net, rays.to(dev), _noise={k: v.to(dev) for k, v in noise.items()})
    assert O.psnr(out.fine.rgb.cpu(), torch.from_numpy(g["fine_rgb"])) >= 85.0
    g2, scene2, _, _, _, rays2, noise2 = golden_setup("srn_mini_64_128")
    net2 = build_net(dev, scene2, precision="f16x3")
    assert net2.packed(True).precision == _lib.PREC_F16X3 and net2.tables(True).dtype == torch.float32
    with torch.no_grad():
        out2 = rend(net2, rays2.to(dev), _noise={k: v.to(dev) for k, v in noise2.items()})
    assert O.psnr(out2.fine.rgb.cpu(), torch.from_numpy(g2["fine_rgb"])) >= 85.0
    sc = dscene(ops, dev, "sn64")
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    with pytest.raises(_lib.PixelNerfHipError):
        ops.eval_points(sc, ops.pack_mlp(state, "f16x3"), torch.zeros(1, 8, 3, device=dev), torch.ones(1, 8, 3, device=dev),
                        tables=ops.fold_latent(sc, state, "f16"))


def test_split_operands_saturate_instead_of_overflowing(ops, dev):
    """hidden activations far beyond the fp16 range: heads clamp at 65504 and tails at f16(v - 65504) under MODE.FP16_OVFL
    (the tail comes out of v_fma_mix{lo,hi}_f16), so every output stays finite -- no inf * 0 -> NaN in the accumulators."""
    s, _ = scene_for("sn64")
    sc = dscene(ops, dev, "sn64")
    p = {k: v.clone() for k, v in mlp_params(11).items()}
    p["lin_z.0.weight"] *= 1e6
    state = {k: v.to(dev) for k, v in p.items()}
    g = load_golden("stages")
    xyz = torch.from_numpy(g["sn64_xyz"]).to(dev)
    vd = torch.from_numpy(g["sn64_viewdirs"]).to(dev)
    out = ops.eval_points(sc, ops.pack_mlp(state, "f16x3"), xyz, vd, tables=ops.fold_latent(sc, state, "f16x3"))
    assert torch.isfinite(out).all()


# ---------------------------------------------------------------- fp16-range guard (pnr_saturation_guard)
def _guarded_eval(ops, dev, state, name="sn64", coarse_slot=0, fold_under_guard=False):
    s, _ = scene_for(name)
    sc = dscene(ops, dev, name)
    g = load_golden("stages")
    xyz = torch.from_numpy(g[name + "_xyz"]).to(dev)
    vd = torch.from_numpy(g[name + "_viewdirs"]).to(dev)
    pk, tab = ops.pack_mlp(state, "f16x3"), ops.fold_latent(sc, state, "f16x3")
    plain = ops.eval_points(sc, pk, xyz, vd, tables=tab).clone()
    ops.saturation_guard_arm(dev)
    try:
        ops.saturation_guard_slot(dev, coarse_slot)
        if fold_under_guard:
            tab = ops.fold_latent(sc, state, "f16x3")
        guarded = ops.eval_points(sc, pk, xyz, vd, tables=tab).clone()
    finally:
        ops.saturation_guard_disarm(dev)
    bits = ops.saturation_guard_poll(dev, wait=True)
    assert ops.saturation_guard_poll(dev) is None  # consumed
    return plain, guarded, bits


@pytest.mark.parametrize("name", ["sn64", "mv_mini"])
def test_saturation_guard_is_silent_on_in_range_networks_and_changes_no_bit(ops, dev, name):
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    plain, guarded, bits = _guarded_eval(ops, dev, state, name)
    assert bits == (0, 0)
    print(f"guarded vs plain instantiation [{name}]: max abs diff {float((plain - guarded).abs().max()):.3e}")
    assert torch.equal(plain, guarded)  # the guarded instantiation computes the same bits


def test_saturation_guard_names_the_layer_that_left_the_fp16_range(ops, dev):
    """lin_z[0] scaled by 1e6: the stream entering block 0 is far beyond 65504 -> bit 0 (relu(x) entering blocks.0.fc_0) and
    everything downstream; a hot lin_out input only -> bit 10 alone; the fine-network slot reports into the second word"""
    p = {k: v.clone() for k, v in mlp_params(11).items()}
    p["lin_z.0.weight"] *= 1e5  # weights ~6e3 (inside the fp16 range: the fold is clean), table entries ~1e5 (beyond it)
    _, out, bits = _guarded_eval(ops, dev, {k: v.to(dev) for k, v in p.items()})
    assert torch.isfinite(out).all()
    assert bits[0] & 1 and bits[1] == 0 and not bits[0] >> 12 & 1, bits
    assert "blocks.0.fc_0" in ops.describe_saturation(bits[0])
    # a lin_z weight itself beyond the range: the per-texel fold (run while the guard is armed) reports it (bit 12)
    p = {k: v.clone() for k, v in mlp_params(11).items()}
    p["lin_z.2.weight"] *= 1e7
    _, out, bits = _guarded_eval(ops, dev, {k: v.to(dev) for k, v in p.items()}, fold_under_guard=True)
    assert torch.isfinite(out).all() and bits[0] >> 12 & 1, bits
    # only the last residual update is large: fc_1 of block 4 scaled up -> the stream in front of lin_out saturates, nothing before it
    p = {k: v.clone() for k, v in mlp_params(11).items()}
    p["blocks.4.fc_1.weight"] *= 3e5
    _, out, bits = _guarded_eval(ops, dev, {k: v.to(dev) for k, v in p.items()}, coarse_slot=1)
    assert bits[0] == 0 and bits[1] == 1 << 10, bits
    assert ops.describe_saturation(bits[1]) == "the stream in front of lin_out"


def test_renderer_warns_once_when_a_checkpoint_leaves_the_fp16_range(dev):
    """API level: every inference render runs guarded (round 6 default); a verdict is reported by the NEXT call as a
    RuntimeWarning that names network and layer.  In-range weights: no warning, same bits call after call."""
    import warnings
    from pixelnerf_amd.render import NeRFRenderer
    from test_api_gpu import build_net
    g, scene, meta, mc, mf, rays, noise = golden_setup("sn64_64_128")
    net = build_net(dev, scene, precision="f16x3")
    rend = NeRFRenderer(n_coarse=64, n_fine=128, n_fine_depth=16, white_bkgd=True).to(dev).eval()
    nz = {k: v.to(dev) for k, v in noise.items()}
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)  # a guard report would fail the in-range part
        a = rend(net, rays.to(dev), _noise=nz)
        torch.cuda.synchronize()
        b = rend(net, rays.to(dev), _noise=nz)
        assert torch.equal(a.fine.rgb, b.fine.rgb)
        assert net.__dict__["_guard_calls"] == 2
        assert net._guard_report(wait=True) == (0, 0)  # both calls ran guarded and both verdicts are clean
    with torch.no_grad():
        net.mlp_fine.lin_z[1].weight.mul_(1e5)  # only the FINE network leaves the range, at block 1 (the weights themselves stay inside)
        rend(net, rays.to(dev), _noise=nz)      # guarded: new weights
        torch.cuda.synchronize()
        with pytest.warns(RuntimeWarning, match=r"fine network: .*blocks\.1\.fc_0"):
            rend(net, rays.to(dev), _noise=nz)


def _border_row_scene(dev):
    """sn64 with the top and bottom texel rows of the encoded grid at 3e4 in every channel (inside the fp16 range: the per-texel fold
    is clean; lin_z[0] of such a texel is ~N(0, 4e4^2) per hidden feature: dozens of features beyond 65504): the centre rays of the target view project onto rows
    7..23 of the 32 only (v = 14.4 .. 45.5 px of 64), the same rays from an origin shifted by (40, 40, 40) onto v = 68..69 px --
    below the source image, i.e. border-clamped onto the bottom row.  -> (scene, in-range rays, saturating rays), (1, 256, 8) each"""
    from testdata import synthetic
    scene, meta = synthetic.make_scene("sn64")
    scene = dict(scene)
    lat = scene["latent"].clone()
    lat[:, :, 0, :] = 3e4
    lat[:, :, -1, :] = 3e4
    scene["latent"] = lat
    inside = synthetic.target_rays(meta).reshape(64, 64, 8)[24:40, 24:40].reshape(1, -1, 8).contiguous()
    outside = inside.clone()
    outside[..., :3] += 40.0
    return scene, inside.to(dev), outside.to(dev)


@pytest.mark.parametrize("mode", ["default", "sample"])
def test_second_batch_that_alone_saturates_is_reported(dev, monkeypatch, mode):
    """saturation depends on the QUERY POINTS too.  Same weights, same encoded scene; the first ray batch
    stays in range, the second one alone looks up texels that drive block 0 past 65504.  Default policy (every inference call
    guarded): the second batch is reported.  PIXELNERF_SATURATION_GUARD=sample (the round-5 policy, kept as the opt-out): only the
    first call after encode() is guarded and the second batch's saturated render goes unreported -- pinned here as the
    documented behaviour of the opt-out."""
    import warnings
    from pixelnerf_amd.render import NeRFRenderer
    from test_api_gpu import build_net
    if mode == "sample":
        monkeypatch.setenv("PIXELNERF_SATURATION_GUARD", "sample")
    else:
        monkeypatch.delenv("PIXELNERF_SATURATION_GUARD", raising=False)
    scene, inside, outside = _border_row_scene(dev)
    net = build_net(dev, scene, precision="f16x3")
    rend = NeRFRenderer(n_coarse=64, n_fine=128, n_fine_depth=16, white_bkgd=True).to(dev).eval()
    with torch.no_grad():
        with warnings.catch_warnings():
            warnings.simplefilter("error", RuntimeWarning)
            rend(net, inside)                        # first batch after "encode": in range
            assert net._guard_report(wait=True) == (0, 0)
        out = rend(net, outside)                     # second batch: border-clamped onto the 3e4 rows
        assert torch.isfinite(out.fine.rgb).all()    # (heads saturate, nothing overflows)
        if mode == "default":
            with pytest.warns(RuntimeWarning, match=r"coarse network: .*blocks\.0\.fc_0"):
                net._guard_report(wait=True)
            rend(net, outside)                       # ... or by the next call, like any verdict
            torch.cuda.synchronize()
            with pytest.warns(RuntimeWarning, match=r"65504"):
                rend(net, inside)
        else:
            with warnings.catch_warnings():
                warnings.simplefilter("error", RuntimeWarning)
                assert net._guard_report(wait=True) is None  # nothing ran guarded: nothing to report


# ---------------------------------------------------------------- combine_type = "max" (util.py:467-468)
@pytest.mark.parametrize("scene_name", MV_STAGE_SCENES)
@pytest.mark.parametrize("precision,bar_rgb,bar_sigma", [("f16x3", 2e-5, 1e-4), ("f32", 2e-5, 1e-4), ("f16", 6e-3, 2e-2)])
def test_view_maximum_matches_reference(ops, dev, scene_name, precision, bar_rgb, bar_sigma):
    """the reference with both ResnetFCs at combine_type "max" (tests/golden/combine_max.npz, frozen from the unmodified reference):
    the flag word of the packed network switches the pooled tile of every inference kernel from the view mean to the view maximum"""
    g, gm = load_golden("stages"), load_golden("combine_max")
    sc = dscene(ops, dev, scene_name)
    xyz = torch.from_numpy(g[f"{scene_name}_xyz"]).to(dev)
    vd = torch.from_numpy(g[f"{scene_name}_viewdirs"]).to(dev)
    for which, seed in (("coarse", 11), ("fine", 12)):
        state = {k: v.to(dev) for k, v in mlp_params(seed).items()}
        if precision == "f32":
            out = ops.eval_points(sc, ops.pack_mlp(state, "f32", combine_max=True), xyz, vd)
        else:
            out = ops.eval_points(sc, ops.pack_mlp(state, precision, folded=True, combine_max=True), xyz, vd,
                                  tables=ops.fold_latent(sc, state, precision))
        out = out.cpu().numpy()
        ref = gm[f"{scene_name}_out_{which}"]
        e_rgb = np.abs(out[..., :3] - ref[..., :3]).max()
        e_s = (np.abs(out[..., 3] - ref[..., 3]) / np.maximum(1.0, ref[..., 3])).max()
        print(f"view maximum {scene_name} {which} {precision}: rgb max err {e_rgb:.3e}, sigma rel err {e_s:.3e}")
        assert e_rgb <= bar_rgb and e_s <= bar_sigma
        assert np.abs(out - g[f"{scene_name}_out_{which}"]).max() > 0.5  # not the view mean


def test_view_maximum_through_the_model_api_and_training_guard(dev):
    """make_model(conf) with mlp.combine_type = "max": renders through NeRFRenderer equal the oracle's; training raises"""
    from pixelnerf_amd.render import NeRFRenderer
    from test_api_gpu import build_net
    g, scene, meta, mc, mf, rays, noise = golden_setup("srn_mini_64_128")
    net = build_net(dev, scene, precision="f16x3")
    net.mlp_coarse.combine_type = net.mlp_fine.combine_type = "max"
    rend = NeRFRenderer(n_coarse=64, n_fine=128, n_fine_depth=16, white_bkgd=True).to(dev).eval()
    nz = {k: v.to(dev) for k, v in noise.items()}
    with torch.no_grad():
        out = rend(net, rays.to(dev), _noise=nz)
    R = rays.shape[0] * rays.shape[1]
    z_c = O.sample_coarse(rays.reshape(-1, 8), noise["u1"], 64)
    pts = (rays.reshape(-1, 8)[:, None, :3] + z_c.unsqueeze(2) * rays.reshape(-1, 8)[:, None, 3:6]).reshape(scene["SB"], -1, 3)
    vdr = rays.reshape(-1, 8)[:, None, 3:6].expand(-1, 64, -1).reshape(scene["SB"], -1, 3)
    with torch.no_grad():
        o = O.pixelnerf_forward(scene, mc, pts, vdr, combine_type="max").reshape(R, 64, 4)
    _, rgb_ref, _ = O.composite_from_rgbsigma(rays.reshape(-1, 8), z_c, o, True)
    assert (out.coarse.rgb.cpu().reshape(-1, 3) - rgb_ref).abs().max() <= 2e-5
    assert (out.coarse.rgb.cpu().reshape(-1, 3) - torch.from_numpy(g["coarse_rgb"]).reshape(-1, 3)).abs().max() > 1e-2  # not the mean
    net.train()
    net.mlp_coarse.lin_in.weight.requires_grad_(True)
    with pytest.raises(NotImplementedError, match="combine_type='max'"):
        rend.train()(net, rays.to(dev), _noise=nz)


def test_parameters_viewed_at_a_4_byte_offset_into_a_flat_buffer(ops, dev):
    """ONE alignment contract for every kernel that reads PnrMlpWeights pointers.  The parameters are caller
    memory: views into a flat buffer (`flat[1:]`: 4-byte aligned, not 16) must give the bits of separately allocated tensors through
    the packers (f16x3 and f16 streams), the per-texel fold, the exact-fp32 GEMM path, the content checksum behind the packed-weight
    cache and the fp32-class training backward."""
    import ctypes
    p = mlp_params(11)
    flat = torch.empty(sum(v.numel() for v in p.values()) + 1, dtype=torch.float32, device=dev)
    off, mis = 1, {}
    for k, v in p.items():
        mis[k] = flat[off:off + v.numel()].view(v.shape)
        mis[k].copy_(v)
        off += v.numel()
    ali = {k: v.to(dev) for k, v in p.items()}
    assert all(t.data_ptr() % 16 != 0 for t in list(mis.values())[:1]) and all(t.data_ptr() % 4 == 0 for t in mis.values())
    sc = dscene(ops, dev, "mv_mini")
    g = load_golden("stages")
    xyz, vd = torch.from_numpy(g["mv_mini_xyz"]).to(dev), torch.from_numpy(g["mv_mini_viewdirs"]).to(dev)
    for prec in ("f16x3", "f16", "f32"):
        outs = []
        for state in (ali, mis):
            if prec == "f32":
                outs.append(ops.eval_points(sc, ops.pack_mlp(state, "f32"), xyz, vd))
            else:
                outs.append(ops.eval_points(sc, ops.pack_mlp(state, prec, folded=True), xyz, vd, tables=ops.fold_latent(sc, state, prec)))
        assert torch.equal(outs[0], outs[1]), prec
    # the content fingerprint (pnr_params_checksum): same parameters, same 64-bit sum, wherever they live
    lib = ops._lib.load()
    sums = []
    for state in (ali, mis):
        w, keep = ops._weights_struct(state)
        ws = torch.zeros(lib.pnr_params_checksum_ws_bytes() // 8, dtype=torch.int64, device=dev)
        out = torch.zeros(1, dtype=torch.int64, device=dev)
        ops._lib.check(lib.pnr_params_checksum(ctypes.byref(w), ops._p(ws), ops._p(out), None, None, ops._stream()), "pnr_params_checksum")
        sums.append(int(out.item()))
    assert sums[0] == sums[1] and sums[0] != 0
    # fp32-class training backward through the C ABI's weight struct (pnr_mlp_backward_split reads the raw parameters)
    s, meta = scene_for("mv_mini")
    from testdata import synthetic
    rays = synthetic.target_rays(meta, n_rays=24).reshape(-1, 8).to(dev)
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], 32, generator=torch.Generator().manual_seed(5)).to(dev))
    gout = torch.randn(rays.shape[0], 32, 4, generator=torch.Generator().manual_seed(6)).to(dev) * 1e-3
    grads = []
    for state in (ali, mis):
        pk, tab = ops.pack_mlp(state, "f16x3"), ops.fold_latent(sc, state, "f16x3")
        out, saved = ops.eval_ray_samples_split_train(sc, pk, tab, rays, z)
        gr, d_zlat, _ = ops.mlp_backward_split(ops.pack_mlp(state, "f32"), saved, gout.reshape(-1, 4))
        grads.append((out, gr, d_zlat))
    assert torch.equal(grads[0][0], grads[1][0]) and torch.equal(grads[0][2], grads[1][2])
    for k in grads[0][1]:
        assert torch.equal(grads[0][1][k], grads[1][1][k]), k
