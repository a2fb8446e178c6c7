"""
GPU parity tests (-m gpu) of the folded inference form (ops.fold_latent + folded streams): lin_z[b] applied to the
encoded grid once per scene (per-texel 16-bit tables) instead of once per sample.  lin_z(bilinear(grid)) ==
bilinear(lin_z(grid)) exactly in real arithmetic; numerically the folded path is held to the SAME tolerances as the
unfolded fused kernel (tests/test_hip_parity.py): f16 per-point |rgb| <= 6e-3 max / 6e-4 mean, render PSNR >= 52 dB;
bf16 5e-2 / 5e-3, >= 36 dB -- against the reference goldens.
"""
import numpy as np
import pytest
import torch

from helpers import RENDER_SCENARIOS, golden_setup, load_golden, mlp_params, scene_for, STAGE_SCENES
from oracle import pnr_oracle as O
from test_hip_parity import PREC_TOL

pytestmark = pytest.mark.gpu


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


def folded(ops, dev, sc, seed, prec):
    state = {k: v.to(dev) for k, v in mlp_params(seed).items()}
    return ops.pack_mlp(state, prec, folded=True), ops.fold_latent(sc, state, prec)


def test_tables_are_lin_z_of_the_grid(ops, dev):
    """table[b][texel][slot_of(f)] == (W_z[b] grid[texel] + b_z[b])[f] (fp32 MFMA, then one f16 rounding)."""
    s, _ = scene_for("mv_mini")
    sc = dscene(ops, dev, "mv_mini")
    p = mlp_params(12)
    tab = ops.fold_latent(sc, {k: v.to(dev) for k, v in p.items()}, "f16").float().cpu()  # (3, NV, Hl, Wl, 512)
    perm = ops.storage_perm().long()  # storage position e -> feature
    grid = s["latent"].permute(0, 2, 3, 1)  # NHWC
    for b in range(3):
        ref = grid @ p[f"lin_z.{b}.weight"].t() + p[f"lin_z.{b}.bias"]  # (..., 512) feature order
        got = torch.empty_like(ref)
        got[..., perm] = tab[b]
        assert (got - ref).abs().max() <= 1e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("scene_name", STAGE_SCENES)
def test_folded_eval_points_matches_reference(ops, dev, scene_name, prec):
    g = load_golden("stages")
    tol = PREC_TOL[prec]
    sc = dscene(ops, dev, scene_name)
    xyz = torch.from_numpy(g[f"{scene_name}_xyz"]).to(dev)
    vd = torch.from_numpy(g[f"{scene_name}_viewdirs"]).to(dev)
    for which, seed in (("coarse", 11), ("fine", 12)):
        pk, tab = folded(ops, dev, sc, seed, prec)
        out = ops.eval_points(sc, pk, xyz, vd, tables=tab).cpu().numpy()
        ref = g[f"{scene_name}_out_{which}"]
        e_rgb = np.abs(out[..., :3] - ref[..., :3])
        assert np.isfinite(out).all()
        assert e_rgb.max() <= tol["rgb_max"], f"rgb max err {e_rgb.max():.3e}"
        assert e_rgb.mean() <= tol["rgb_mean"], f"rgb mean err {e_rgb.mean():.3e}"
        e_s = np.abs(out[..., 3] - ref[..., 3]) / np.maximum(1.0, ref[..., 3])
        assert e_s.max() <= tol["sigma_rel"], f"sigma rel err {e_s.max():.3e}"


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("name", RENDER_SCENARIOS)
def test_folded_render_forward_matches_reference(ops, dev, name, prec):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    tol = PREC_TOL[prec]
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    sc = dscene(ops, dev, str(g["scene"]))
    pc, tc = folded(ops, dev, sc, int(g["mlp_seed_coarse"]), prec)
    pf, tf = folded(ops, dev, sc, int(g["mlp_seed_fine"]), prec) if mf is not None else (None, None)
    r = rays.reshape(-1, 8).to(dev)
    out = ops.render_forward(sc, pc, pf, r, Kc, Kf, Kfd, {k: v.to(dev) for k, v in noise.items()},
                             depth_std=float(g["depth_std"]), white_bkgd=bool(g["white_bkgd"]),
                             lindisp=bool(g["lindisp"]), want_weights=True, tables=(tc, tf))
    span = float(meta["z_far"] - meta["z_near"])
    for p in ["coarse"] + (["fine"] if Kf > 0 else []):
        rgb = out[p]["rgb"].cpu()
        depth = out[p]["depth"].cpu().numpy()
        assert np.isfinite(rgb.numpy()).all() and np.isfinite(depth).all()
        ps = O.psnr(rgb, torch.from_numpy(g[f"{p}_rgb"]).reshape(-1, 3))
        assert ps >= tol["psnr"], f"{p} PSNR {ps:.1f} dB"
        ed = np.abs(depth - g[f"{p}_depth"].reshape(-1))
        assert np.percentile(ed, 99) <= tol["depth_p99"] * span


def test_folded_and_unfolded_agree_and_variants_match(ops, dev):
    """Same network, both forms, 64x64 view at 64 samples: per-point outputs within the f16 band of each other;
    ray-sample and explicit-point variants of the folded kernel agree bitwise; deterministic."""
    from testdata import synthetic
    s, meta = scene_for("sn64")
    sc = dscene(ops, dev, "sn64")
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    pk_u = ops.pack_mlp(state, "f16")
    pk_f, tab = ops.pack_mlp(state, "f16", folded=True), ops.fold_latent(sc, state, "f16")
    rays = synthetic.target_rays(meta).reshape(-1, 8).to(dev)
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], 64, device=dev))
    a = ops.eval_ray_samples(sc, pk_u, rays, z)
    b = ops.eval_ray_samples(sc, pk_f, rays, z, tables=tab)
    assert (a[..., :3] - b[..., :3]).abs().max().item() <= 6e-3
    assert torch.equal(b, ops.eval_ray_samples(sc, pk_f, rays, z, tables=tab))
    pts = (rays[:, None, :3] + z.unsqueeze(2) * rays[:, None, 3:6]).reshape(1, -1, 3)
    vd = rays[:, None, 3:6].expand(-1, 64, -1).reshape(1, -1, 3)
    c = ops.eval_points(sc, pk_f, pts.contiguous(), vd.contiguous(), tables=tab).reshape(b.shape)
    assert torch.equal(b, c)


def test_folded_api_errors_and_saturation(ops, dev):
    from pixelnerf_amd import _lib
    sc = dscene(ops, dev, "sn64")
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    pk_u, pk_f, tab = ops.pack_mlp(state, "f16"), ops.pack_mlp(state, "f16", folded=True), ops.fold_latent(sc, state, "f16")
    g = load_golden("stages")
    xyz, vd = torch.from_numpy(g["sn64_xyz"]).to(dev), torch.from_numpy(g["sn64_viewdirs"]).to(dev)
    with pytest.raises(_lib.PixelNerfHipError):
        ops.eval_points(sc, pk_f, xyz, vd)                # folded stream without tables
    with pytest.raises(_lib.PixelNerfHipError):
        ops.eval_points(sc, pk_u, xyz, vd, tables=tab)    # full stream with tables
    with pytest.raises(_lib.PixelNerfHipError):
        ops.fold_latent(sc, state, "f32")
    # table entries beyond the f16 range saturate instead of turning into inf
    big = {k: v.clone() for k, v in state.items()}
    big["lin_z.0.weight"] *= 1e6
    tab_big = ops.fold_latent(sc, big, "f16")
    assert torch.isfinite(tab_big.float()).all()
    out = ops.eval_points(sc, ops.pack_mlp(big, "f16", folded=True), xyz, vd, tables=tab_big)
    assert torch.isfinite(out).all()


def test_tile_size_does_not_change_the_result(ops, dev):
    """The folded single-view kernel runs 96-point tiles for long launches and 64-point tiles when that takes fewer
    rounds (pnr_mlp.hip: use_tile96): 98 304 points in one launch (96-point tiles) against the same points in chunks of
    8 192 (64-point tiles) -- a point's output does not depend on the tile it sits in, bit for bit."""
    from testdata import synthetic
    s, meta = scene_for("sn64")
    sc = dscene(ops, dev, "sn64")
    state = {k: v.to(dev) for k, v in mlp_params(12).items()}
    pk, tab = ops.pack_mlp(state, "f16", folded=True), ops.fold_latent(sc, state, "f16")
    rays = synthetic.target_rays(meta).reshape(-1, 8)[:1536].contiguous().to(dev)  # 1536 rays x 64 samples
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], 64, device=dev, generator=torch.Generator(device=dev).manual_seed(3)))
    whole = ops.eval_ray_samples(sc, pk, rays, z, tables=tab)
    parts = [ops.eval_ray_samples(sc, pk, rays[i:i + 128].contiguous(), z[i:i + 128].contiguous(), tables=tab)
             for i in range(0, rays.shape[0], 128)]
    assert torch.equal(whole, torch.cat(parts, dim=0))
