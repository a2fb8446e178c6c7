"""
GPU parity tests (-m gpu): the HIP path, called through the C ABI (pixelnerf_amd.ops ->
libpixelnerf_hip.so), against the CPU oracle and the golden vectors frozen from the
reference.

Stated tolerances (the reference computes everything in fp32):
  * sampling / compositing / ray generation are fp32 on both sides: z within 1e-6*(far-near)
    [2e-6 for lindisp], weights/rgb within 2e-6 .. 1e-5, exact importance-bin indices except
    for a <=0.2 % allowance for cdf ties at rounding level;
  * the fused network runs its 512-wide linears on the matrix cores with 16-bit operands and
    fp32 accumulation.  Per-point outputs vs the fp32 oracle:
        f16 : |rgb| err <= 6e-3 max, <= 6e-4 mean;  sigma err <= 2e-2 * max(1, sigma)
        bf16: |rgb| err <= 5e-2 max, <= 5e-3 mean;  sigma err <= 1.5e-1 * max(1, sigma)
  * end-to-end renders (identical rays, weights, grid and noise) vs the reference goldens:
        f16 : PSNR >= 52 dB, depth |err| p99 <= 5e-3*(far-near)
        bf16: PSNR >= 36 dB, depth |err| p99 <= 3e-2*(far-near)
"""
import numpy as np
import pytest
import torch

from helpers import RENDER_SCENARIOS, assert_close_frac, golden_setup, load_golden, mlp_params, scene_for, STAGE_SCENES
from oracle import pnr_oracle as O

pytestmark = pytest.mark.gpu

PREC_TOL = {
    # per-point: rgb max, rgb mean, sigma rel ; render: psnr, depth p99 (fraction of span)
    "f16": dict(rgb_max=6e-3, rgb_mean=6e-4, sigma_rel=2e-2, psnr=52.0, depth_p99=5e-3),
    "bf16": dict(rgb_max=5e-2, rgb_mean=5e-3, sigma_rel=1.5e-1, psnr=36.0, depth_p99=3e-2),
}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from pixelnerf_amd import ops as _ops
    return _ops


_scene_cache = {}
_pack_cache = {}


def dscene(ops, dev, name):
    if name not in _scene_cache:
        s, _ = scene_for(name)
        _scene_cache[name] = ops.make_scene(s["latent"].to(dev), s["poses"].to(dev), s["focal"].to(dev),
                                            s["c"].to(dev), s["image_shape"], s["NS"])
    return _scene_cache[name]


def packed(ops, dev, seed, prec):
    if (seed, prec) not in _pack_cache:
        _pack_cache[(seed, prec)] = ops.pack_mlp({k: v.to(dev) for k, v in mlp_params(seed).items()}, prec)
    return _pack_cache[(seed, prec)]


# ------------------------------------------------------------------ fp32 stages


@pytest.mark.parametrize("name", RENDER_SCENARIOS)
def test_sample_coarse_matches_oracle(ops, dev, name):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    r = rays.reshape(-1, 8)
    z = ops.sample_coarse(r.to(dev), noise["u1"].to(dev), lindisp=bool(g["lindisp"])).cpu()
    span = float(meta["z_far"] - meta["z_near"])
    np.testing.assert_allclose(z.numpy(), g["coarse_z"], rtol=0, atol=2e-6 * span)


@pytest.mark.parametrize("name", RENDER_SCENARIOS)
def test_composite_matches_reference(ops, dev, name):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    r = rays.reshape(-1, 8).to(dev)
    for p in ["coarse"] + (["fine"] if int(g["n_fine"]) > 0 else []):
        K = g[f"{p}_z"].shape[-1]
        w, rgb, depth = ops.composite(r, torch.from_numpy(g[f"{p}_z"]).to(dev),
                                      torch.from_numpy(g[f"{p}_rgbsigma"]).to(dev),
                                      white_bkgd=bool(g["white_bkgd"]))
        np.testing.assert_allclose(w.cpu().numpy(), g[f"{p}_weights"].reshape(-1, K), rtol=0, atol=2e-6)
        np.testing.assert_allclose(rgb.cpu().numpy(), g[f"{p}_rgb"].reshape(-1, 3), rtol=0, atol=5e-6)
        np.testing.assert_allclose(depth.cpu().numpy(), g[f"{p}_depth"].reshape(-1), rtol=0, atol=2e-5)
        # without the weights output
        w2, rgb2, _ = ops.composite(r, torch.from_numpy(g[f"{p}_z"]).to(dev),
                                    torch.from_numpy(g[f"{p}_rgbsigma"]).to(dev),
                                    white_bkgd=bool(g["white_bkgd"]), want_weights=False)
        assert w2 is None and torch.equal(rgb2, rgb)


@pytest.mark.parametrize("name", [n for n in RENDER_SCENARIOS if n != "sn64_c32"])
def test_sample_fine_matches_reference(ops, dev, name):
    """importance + depth samples + sort from the reference's own coarse weights / depth."""
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    r = rays.reshape(-1, 8).to(dev)
    span = float(meta["z_far"] - meta["z_near"])
    z = ops.sample_fine(
        r, torch.from_numpy(g["coarse_weights"]).reshape(-1, Kc).to(dev),
        torch.from_numpy(g["coarse_depth"]).reshape(-1).to(dev), torch.from_numpy(g["coarse_z"]).to(dev),
        noise["u2"].to(dev) if Kf - Kfd > 0 else None, noise["u3"].to(dev) if Kf - Kfd > 0 else None,
        noise["n4"].to(dev) if Kfd > 0 else None, depth_std=float(g["depth_std"]), lindisp=bool(g["lindisp"]))
    z = z.cpu().numpy()
    assert z.shape == g["fine_z"].shape
    assert (np.diff(z, axis=1) >= 0).all(), "fine samples must come out sorted"
    assert_close_frac(z, g["fine_z"], 2e-6 * span, max_frac=2e-3, loose_atol=span / Kc * 1.01, what="fine z")


def test_gen_rays_matches_reference_formula(ops, dev):
    from testdata import synthetic
    for name in ("sn64", "dtu_mini"):
        _, meta = scene_for(name)
        poses = torch.stack([meta["pre"] @ synthetic.pose_spherical(t, -20.0, meta["radius"]) for t in (10.0, 75.0)])
        ref = synthetic.gen_rays(poses, meta["W"], meta["H"], meta["focal"], meta["z_near"], meta["z_far"], c=meta["c"])
        out = ops.gen_rays(poses.to(dev), meta["W"], meta["H"], meta["focal"], meta["z_near"], meta["z_far"], c=meta["c"])
        np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-6)


def test_nchw_to_nhwc(ops, dev):
    x = torch.randn(3, 512, 15, 20, device=dev)
    y = ops.nchw_to_nhwc(x)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())


# ------------------------------------------------------------------ fused network


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("scene_name", STAGE_SCENES)
def test_eval_points_matches_reference(ops, dev, scene_name, prec):
    """net(xyz, coarse=, viewdirs=) against PixelNeRFNet.forward of the reference (goldens)."""
    g = load_golden("stages")
    tol = PREC_TOL[prec]
    sc = dscene(ops, dev, scene_name)
    xyz = torch.from_numpy(g[f"{scene_name}_xyz"]).to(dev)
    vd = torch.from_numpy(g[f"{scene_name}_viewdirs"]).to(dev)
    for which, seed in (("coarse", 11), ("fine", 12)):
        out = ops.eval_points(sc, packed(ops, dev, seed, prec), xyz, vd).cpu().numpy()
        ref = g[f"{scene_name}_out_{which}"]
        e_rgb = np.abs(out[..., :3] - ref[..., :3])
        assert np.isfinite(out).all()
        assert e_rgb.max() <= tol["rgb_max"], f"rgb max err {e_rgb.max():.3e}"
        assert e_rgb.mean() <= tol["rgb_mean"], f"rgb mean err {e_rgb.mean():.3e}"
        e_s = np.abs(out[..., 3] - ref[..., 3]) / np.maximum(1.0, ref[..., 3])
        assert e_s.max() <= tol["sigma_rel"], f"sigma rel err {e_s.max():.3e}"


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_eval_ray_samples_equals_eval_points(ops, dev, prec):
    """Variant A (rays + z, fuses o + z d) and variant B (explicit xyz/viewdirs) agree bitwise."""
    g, scene, meta, mc, mf, rays, noise = golden_setup("train_64_32")
    sc = dscene(ops, dev, "train")
    r = rays.reshape(-1, 8).to(dev)
    z = torch.from_numpy(g["fine_z"]).to(dev)  # (R, 96): tiles straddle rays and objects
    pk = packed(ops, dev, 12, prec)
    a = ops.eval_ray_samples(sc, pk, r, z)
    SB = rays.shape[0]
    pts = (r[:, None, :3] + z.unsqueeze(2) * r[:, None, 3:6]).reshape(SB, -1, 3)
    vd = r[:, None, 3:6].expand(-1, z.shape[1], -1).reshape(SB, -1, 3)
    b = ops.eval_points(sc, pk, pts.contiguous(), vd.contiguous()).reshape(a.shape)
    # o + z*d is rounded identically (no FMA contraction in either path)
    assert torch.equal(a, b)


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("name", RENDER_SCENARIOS)
def test_render_forward_matches_reference(ops, dev, name, prec):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    tol = PREC_TOL[prec]
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    sc = dscene(ops, dev, str(g["scene"]))
    pc = packed(ops, dev, int(g["mlp_seed_coarse"]), prec)
    pf = packed(ops, dev, int(g["mlp_seed_fine"]), prec) if mf is not None else None
    r = rays.reshape(-1, 8).to(dev)
    out = ops.render_forward(sc, pc, pf, r, Kc, Kf, Kfd, {k: v.to(dev) for k, v in noise.items()},
                             depth_std=float(g["depth_std"]), white_bkgd=bool(g["white_bkgd"]),
                             lindisp=bool(g["lindisp"]), want_weights=True)
    span = float(meta["z_far"] - meta["z_near"])
    assert ("fine" in out) == (Kf > 0)
    for p in ["coarse"] + (["fine"] if Kf > 0 else []):
        K = Kc if p == "coarse" else Kc + Kf
        rgb = out[p]["rgb"].cpu()
        depth = out[p]["depth"].cpu().numpy()
        w = out[p]["weights"].cpu().numpy()
        assert rgb.shape == (r.shape[0], 3) and w.shape == (r.shape[0], K)
        assert np.isfinite(rgb.numpy()).all() and np.isfinite(depth).all()
        ps = O.psnr(rgb, torch.from_numpy(g[f"{p}_rgb"]).reshape(-1, 3))
        assert ps >= tol["psnr"], f"{p} PSNR {ps:.1f} dB"
        ed = np.abs(depth - g[f"{p}_depth"].reshape(-1))
        assert np.percentile(ed, 99) <= tol["depth_p99"] * span, f"{p} depth p99 {np.percentile(ed, 99):.3e}"


def test_render_is_deterministic_and_chunk_invariant(ops, dev):
    """Same inputs -> bit-identical outputs (no atomics on the forward path), and rendering
    the rays in two chunks equals rendering them at once (rays are independent units)."""
    g, scene, meta, mc, mf, rays, noise = golden_setup("sn64_64_128")
    sc = dscene(ops, dev, "sn64")
    pc, pf = packed(ops, dev, 11, "f16"), packed(ops, dev, 12, "f16")
    r = rays.reshape(-1, 8).to(dev)
    nz = {k: v.to(dev) for k, v in noise.items()}
    a = ops.render_forward(sc, pc, pf, r, 64, 128, 16, nz, white_bkgd=True)
    b = ops.render_forward(sc, pc, pf, r, 64, 128, 16, nz, white_bkgd=True)
    assert torch.equal(a["fine"]["rgb"], b["fine"]["rgb"]) and torch.equal(a["fine"]["depth"], b["fine"]["depth"])
    h = r.shape[0] // 2 + 5
    parts = [ops.render_forward(sc, pc, pf, r[s].contiguous(), 64, 128, 16,
                                {k: v[s].contiguous() for k, v in nz.items()}, white_bkgd=True)
             for s in (slice(0, h), slice(h, None))]
    cat = torch.cat([p["fine"]["rgb"] for p in parts])
    assert torch.equal(cat, a["fine"]["rgb"])


def test_full_size_properties(ops, dev):
    """BASELINE config (2) at full size (64x64 image, 64+128): size-independent properties."""
    from testdata import synthetic
    scene, meta = scene_for("sn64")
    sc = dscene(ops, dev, "sn64")
    pc, pf = packed(ops, dev, 11, "f16"), packed(ops, dev, 12, "f16")
    rays = synthetic.target_rays(meta).reshape(-1, 8).to(dev)
    R = rays.shape[0]
    assert R == 4096
    nz = {k: v.to(dev) for k, v in synthetic.make_noise(R, 64, 128, 16).items()}
    out = ops.render_forward(sc, pc, pf, rays, 64, 128, 16, nz, white_bkgd=True, want_weights=True)
    for p, K in (("coarse", 64), ("fine", 192)):
        w = out[p]["weights"]
        assert torch.isfinite(w).all() and torch.isfinite(out[p]["rgb"]).all()
        assert (w >= -1e-6).all() and (w.sum(-1) <= 1 + 1e-4).all()  # partition of unity (<= 1)
        d = out[p]["depth"]
        assert (d >= -1e-5).all() and (d <= meta["z_far"] + 1e-4).all()
        # white background: rgb = sum w c + 1 - sum w with c in (0,1)
        assert (out[p]["rgb"] >= -1e-5).all() and (out[p]["rgb"] <= 1 + 1e-4).all()
    # race screen: 16-48 tiles per persistent workgroup, LDS images reused tile after tile, weight
    # ring wrapping across tiles -- a second run must be bit-identical
    out2 = ops.render_forward(sc, pc, pf, rays, 64, 128, 16, nz, white_bkgd=True, want_weights=True)
    for p in ("coarse", "fine"):
        assert torch.equal(out[p]["weights"], out2[p]["weights"]) and torch.equal(out[p]["rgb"], out2[p]["rgb"])
    # and the image must not depend on how the rays are batched (tile/workgroup assignment changes)
    idx = torch.randperm(R, generator=torch.Generator().manual_seed(2)).to(dev)
    out3 = ops.render_forward(sc, pc, pf, rays[idx].contiguous(), 64, 128, 16,
                              {k: v[idx].contiguous() for k, v in nz.items()}, white_bkgd=True)
    assert torch.equal(out3["fine"]["rgb"], out["fine"]["rgb"][idx])


# ------------------------------------------------------------------ edge cases / errors


def test_empty_and_ragged_inputs(ops, dev):
    sc = dscene(ops, dev, "sn64")
    pc = packed(ops, dev, 11, "f16")
    # empty ray batch (reference returns empty tensors, nerf.py:23-27)
    out = ops.render_forward(sc, pc, None, torch.zeros(0, 8, device=dev), 8, 0, 0,
                             {"u1": torch.zeros(0, 8, device=dev)})
    assert out["coarse"]["rgb"].shape == (0, 3)
    # ragged: 5 rays x 7 samples = 35 points (partial tile), and 3 rays x 50 (tiles straddle rays)
    scene, meta = scene_for("sn64")
    from testdata import synthetic
    rays_all = synthetic.target_rays(meta).reshape(-1, 8)
    for R, K in ((5, 7), (3, 50), (1, 1)):
        r = rays_all[100:100 + R].contiguous()
        u = torch.rand(R, K, generator=torch.Generator().manual_seed(3))
        z = O.sample_coarse(r, u, K)
        ref = O.composite(scene, mlp_params(11), r, z, 1, True)
        got = ops.eval_ray_samples(sc, pc, r.to(dev), z.to(dev)).cpu()
        assert np.abs(got[..., :3].numpy() - ref[3][..., :3].numpy()).max() <= PREC_TOL["f16"]["rgb_max"]


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_object_boundary_inside_a_tile(ops, dev, prec):
    """SB=2 x NS=2 with 5 rays x 7 samples per object: 35 points per object, so the object (pose /
    feature-grid) switch happens in the middle of the first 64-point tile, for both variants."""
    from testdata import synthetic
    scene, meta = scene_for("mv_mini")
    sc = dscene(ops, dev, "mv_mini")
    pk = packed(ops, dev, 12, prec)
    rays = synthetic.target_rays(meta, n_rays=5)  # (2, 5, 8)
    r = rays.reshape(-1, 8)
    u = torch.rand(10, 7, generator=torch.Generator().manual_seed(8))
    z = O.sample_coarse(r, u, 7)
    ref = O.composite(scene, mlp_params(12), r, z, 2, True)[3]  # (10, 7, 4)
    got = ops.eval_ray_samples(sc, pk, r.to(dev), z.to(dev)).cpu()
    assert np.abs(got[..., :3].numpy() - ref[..., :3].numpy()).max() <= PREC_TOL[prec]["rgb_max"]
    pts = (r[:, None, :3] + z.unsqueeze(2) * r[:, None, 3:6]).reshape(2, -1, 3)
    vd = r[:, None, 3:6].expand(-1, 7, -1).reshape(2, -1, 3)
    got_b = ops.eval_points(sc, pk, pts.contiguous().to(dev), vd.contiguous().to(dev)).reshape(10, 7, 4)
    assert torch.equal(got_b.cpu(), got)


def test_errors_are_loud(ops, dev):
    from pixelnerf_amd import _lib
    sc = dscene(ops, dev, "sn64")
    pc = packed(ops, dev, 11, "f16")
    with pytest.raises(_lib.PixelNerfHipError):
        ops.sample_coarse(torch.zeros(4, 8), torch.zeros(4, 8))  # CPU tensors: no CPU path
    with pytest.raises(ValueError):
        ops.eval_ray_samples(sc, pc, torch.zeros(4, 7, device=dev), torch.zeros(4, 8, device=dev))
    with pytest.raises(KeyError):
        ops.pack_mlp({"lin_in.weight": torch.zeros(512, 42, device=dev)})
    with pytest.raises(_lib.PixelNerfHipError):  # a ray's cdf + sample set beyond what the LDS holds (2 n_coarse + n_fine >= 10240)
        ops.sample_fine(torch.zeros(2, 8, device=dev), torch.zeros(2, 6000, device=dev), torch.zeros(2, device=dev),
                        torch.zeros(2, 6000, device=dev), torch.zeros(2, 4, device=dev), torch.zeros(2, 4, device=dev), None)


def test_f16_activations_saturate_instead_of_overflowing(ops, dev):
    """Scale the first layers so that hidden activations exceed the fp16 range (65504): the fused
    relu+saturate keeps every output finite (an un-clamped fp16 conversion would give inf -> NaN)."""
    sc = dscene(ops, dev, "sn64")
    p = {k: v.clone() for k, v in mlp_params(11).items()}
    p["lin_z.0.weight"] *= 1e6  # hidden stream ~1e6, far beyond 65504
    pk = ops.pack_mlp({k: v.to(dev) for k, v in p.items()}, "f16")
    g = load_golden("stages")
    out = ops.eval_points(sc, pk, torch.from_numpy(g["sn64_xyz"]).to(dev), torch.from_numpy(g["sn64_viewdirs"]).to(dev))
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("fold", [False, True])
def test_fine_pass_on_the_coarse_network_reuses_coarse_outputs(ops, dev, fold):
    """mlp_fine is None (eval/eval.py:140, models.py:242): the fine pass runs the coarse network; render_forward then
    evaluates only the new samples and merges.  Must be bit-identical to passing the same network explicitly as the
    fine one (which evaluates all Kc+Kf samples)."""
    from testdata import synthetic
    s, meta = scene_for("mv_mini")
    sc = dscene(ops, dev, "mv_mini")
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    pk = ops.pack_mlp(state, "f16", folded=fold)
    tab = ops.fold_latent(sc, state, "f16") if fold else None
    rays = synthetic.target_rays(meta, n_rays=100).reshape(-1, 8).to(dev)
    R = rays.shape[0]
    noise = {k: v.to(dev) for k, v in synthetic.make_noise(R, 24, 40, 8, seed=2).items()}
    kw = dict(white_bkgd=True, want_weights=True)
    a = ops.render_forward(sc, pk, None, rays, 24, 40, 8, noise, tables=None if tab is None else (tab, None), **kw)
    b = ops.render_forward(sc, pk, pk, rays, 24, 40, 8, noise, tables=None if tab is None else (tab, tab), **kw)
    for p in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            assert torch.equal(a[p][k], b[p][k]), (p, k)
