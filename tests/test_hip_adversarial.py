"""
GPU parity tests (-m gpu) on ADVERSARIAL fixtures frozen from the unmodified reference (oracle/make_goldens.py,
VERDICT r01 item 5).  The seeded networks of the other fixtures are "fog" (density > 0 on 97-99 % of the samples,
near-uniform coarse weights); here

  adv_surface_* : testdata.synthetic.surface_variant -- sigma' = relu(100 (s - tau)): 0 on ~90 % of the samples,
                  50..300 on a thin shell, the 4 heaviest coarse bins carry 50-96 % of a ray's weight.  The inverse CDF
                  is a staircase, a 16-bit error in s is amplified 100-fold at the shell's edge; every 4th ray carries an
                  importance draw at the very top of the cdf (u = 1 - 2^-24 -> searchsorted index == n_coarse when
                  cdf[-1] rounds below 1: a fine sample beyond `far`, negative last delta, nerf.py:138-141,181);
  adv_plane     : query points exactly ON a source camera's image plane (u = x/0 = +-inf, 0/0 = NaN) and BEHIND it
                  (mirrored projection), models.py:206-212.  NaN policy, decided by what the reference does: ATen's
                  grid_sample maps a NaN coordinate to 0 (finite output, no NaN propagation); project_point does the same.

Stage-wise checks (reference intermediates in, exact to rounding) pin the discontinuous parts; end-to-end renders go
through helpers.robust_render_stats (rays with a beyond-far sample -- forced rays only -- are ill-conditioned in the
reference itself and are excluded, see its docstring) against ADV_TOL below: the f16 PSNR bar of the fog fixtures holds
unchanged, the depth / bf16 bars are restated with the measured evidence (profiles/r02_adversarial_parity.txt).
"""
import numpy as np
import pytest
import torch

from helpers import ADVERSARIAL_SCENARIOS, assert_close_frac, golden_setup, load_golden, mlp_params, robust_render_stats, scene_for
from test_hip_parity import PREC_TOL

# Render tolerances ON THE ADVERSARIAL SET, restated with evidence (profiles/r02_adversarial_parity.txt, MI355X):
#   f16 : fine PSNR 59-83 dB (fog fixtures: 66-80) -> the 52 dB bar holds; depth p99 1.7e-3..9e-3 of the span
#         (fog: <= 1e-3; bar 5e-3) -> 1.5e-2 here: a 100x density gain moves the shell edge by a bin where s ~ tau;
#   bf16: fine PSNR 33-63 dB (fog ~50; bar 36) -> 30 here, depth p99 up to 0.12 of the span -> 0.15: bf16 operands are
#         not the recommended form for surface-like densities (f16 is the default), and this is the evidence.
ADV_TOL = {"f16": dict(psnr=52.0, depth_p99=1.5e-2), "bf16": dict(psnr=30.0, depth_p99=0.15)}

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


# ------------------------------------------------------------------ stage-wise: exact on the reference's intermediates


@pytest.mark.parametrize("name", ADVERSARIAL_SCENARIOS)
def test_sample_fine_on_peaked_weights_matches_reference(ops, dev, name):
    """inverse-CDF sampling from the reference's own PEAKED coarse weights.  Every sample must land where the reference
    put it (2e-6 of the span, no exceptions) -- except the forced draw u = 1 - 2^-24 of every 4th ray: whether it
    selects index n_coarse (a sample beyond `far`) or the last bin is `u >= cdf[-1]` with cdf[-1] = 1 +- 1 ulp, and
    cdf[-1] inherits the summation order of `torch.sum` over the 64 weights (ATen's vectorised cascade sum on the
    generating CPU vs a wavefront butterfly here; the reference on another CPU or on a GPU differs the same way).
    For that one sample either alternative is accepted; all other samples of those rays must still match."""
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc = int(g["n_coarse"])
    r = rays.reshape(-1, 8).to(dev)
    span = float(meta["z_far"] - meta["z_near"])
    z = ops.sample_fine(r, torch.from_numpy(g["coarse_weights"]).reshape(-1, Kc).to(dev),
                        torch.from_numpy(g["coarse_depth"]).reshape(-1).to(dev), torch.from_numpy(g["coarse_z"]).to(dev),
                        noise["u2"].to(dev), noise["u3"].to(dev), noise["n4"].to(dev), depth_std=float(g["depth_std"]),
                        lindisp=bool(g["lindisp"])).cpu().numpy()
    zg = g["fine_z"]
    near, far = g["rays"].reshape(-1, 8)[:, 6], g["rays"].reshape(-1, 8)[:, 7]
    assert (zg[:, -1] > far).sum() >= 3, "fixture must contain beyond-far samples"
    assert (np.diff(z, axis=1) >= 0).all()
    forced = np.arange(z.shape[0]) % 4 == 0
    assert_close_frac(z[~forced], zg[~forced], 2e-6 * span, max_frac=0.0, what="fine z on peaked weights")
    assert ((z[~forced, -1] > far[~forced]) == (zg[~forced, -1] > far[~forced])).all()
    u3 = noise["u3"][:, 0].numpy()
    n_alt = 0
    for i in np.where(forced)[0]:
        cand = [np.float32(near[i] * (1 - t) + far[i] * t) for t in (np.float32((Kc + u3[i]) / Kc), np.float32((Kc - 1 + u3[i]) / Kc))]

        def without_forced(row):
            j = int(np.argmin(np.minimum(np.abs(row - cand[0]), np.abs(row - cand[1]))))
            return np.delete(row, j), row[j]
        a, fa = without_forced(z[i])
        b, fb = without_forced(zg[i])
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * span)
        assert min(abs(fa - cand[0]), abs(fa - cand[1])) <= 2e-6 * span and min(abs(fb - cand[0]), abs(fb - cand[1])) <= 2e-6 * span
        n_alt += int(abs(fa - fb) > 2e-6 * span)
    print(f"ADVSTAGE {name}: {int(forced.sum())} forced rays, {n_alt} take the other alternative for the top-of-cdf draw")


@pytest.mark.parametrize("name", ADVERSARIAL_SCENARIOS)
def test_composite_with_negative_last_delta_matches_reference(ops, dev, name):
    """compositing of the reference's own samples: includes rays whose last delta is negative (alpha < 0)."""
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    r = rays.reshape(-1, 8).to(dev)
    K = g["fine_z"].shape[-1]
    w, rgb, depth = ops.composite(r, torch.from_numpy(g["fine_z"]).to(dev), torch.from_numpy(g["fine_rgbsigma"]).to(dev),
                                  white_bkgd=bool(g["white_bkgd"]))
    ref_w = g["fine_weights"].reshape(-1, K)
    scale = max(1.0, float(np.abs(ref_w).max()))  # |alpha| >> 1 on ill-conditioned rays: relative to the largest weight
    np.testing.assert_allclose(w.cpu().numpy(), ref_w, rtol=0, atol=5e-6 * scale)
    np.testing.assert_allclose(rgb.cpu().numpy(), g["fine_rgb"].reshape(-1, 3), rtol=0, atol=2e-5 * scale)


@pytest.mark.parametrize("prec", ["f16x3", "f16", "bf16", "f32"])
def test_points_on_and_behind_the_camera_plane(ops, dev, prec):
    g = load_golden("adv_plane")
    sc = dscene(ops, dev, "plane_mini")
    xyz, vd = torch.from_numpy(g["xyz"]).to(dev), torch.from_numpy(g["viewdirs"]).to(dev)
    for which, seed in (("coarse", 11), ("fine", 12)):
        state = {k: v.to(dev) for k, v in mlp_params(seed).items()}
        ref = g[f"out_{which}"]
        if prec == "f16x3":  # the default precision: always folded (fp32 tables), held to the exact path's bar
            forms = [(ops.pack_mlp(state, prec), ops.fold_latent(sc, state, prec))]
        else:
            forms = [(ops.pack_mlp(state, prec), None)]
            if prec != "f32":
                forms.append((ops.pack_mlp(state, prec, folded=True), ops.fold_latent(sc, state, prec)))
        for pk, tab in forms:
            out = ops.eval_points(sc, pk, xyz, vd, tables=tab).cpu().numpy()
            assert np.isfinite(out).all(), "the reference's output is finite on these points (NaN coordinate -> texel 0)"
            e = np.abs(out[..., :3] - ref[..., :3])
            if prec in ("f32", "f16x3"):
                assert e.max() <= 2e-5
            else:
                assert e.max() <= PREC_TOL[prec]["rgb_max"] and e.mean() <= PREC_TOL[prec]["rgb_mean"]


# ------------------------------------------------------------------ end to end


def _render(ops, dev, name, prec, fold):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    sc = dscene(ops, dev, str(g["scene"]))
    st = [{k: v.to(dev) for k, v in m.items()} for m in ((mc, mf) if mf is not None else (mc,))]
    pk = [ops.pack_mlp(s, prec, folded=fold) for s in st]
    tabs = None
    if fold:
        tabs = tuple(ops.fold_latent(sc, s, prec) for s in st) + ((None,) if mf is None else ())
    r = rays.reshape(-1, 8).to(dev)
    nz = {k: v.to(dev) for k, v in noise.items()}
    out = ops.render_forward(sc, pk[0], pk[1] if mf is not None else None, r, Kc, Kf, Kfd, nz, depth_std=float(g["depth_std"]),
                             white_bkgd=bool(g["white_bkgd"]), lindisp=bool(g["lindisp"]), want_weights=True, tables=tabs)
    # the fine samples the call used: same kernels, fed the call's own coarse outputs (deterministic)
    z_c = ops.sample_coarse(r, nz["u1"], bool(g["lindisp"]))
    z_f = ops.sample_fine(r, out["coarse"]["weights"], out["coarse"]["depth"], z_c, nz["u2"], nz["u3"], nz["n4"],
                          depth_std=float(g["depth_std"]), lindisp=bool(g["lindisp"]))
    span = float(meta["z_far"] - meta["z_near"])
    return g, out, z_f, span


@pytest.mark.parametrize("fold", [True, False], ids=["folded", "unfolded"])
@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("name", ADVERSARIAL_SCENARIOS)
def test_surface_like_density_render(ops, dev, name, prec, fold):
    g, out, z_f, span = _render(ops, dev, name, prec, fold)
    tol = ADV_TOL[prec]
    # coarse pass: no discontinuity in front of it
    from oracle import pnr_oracle as O
    ps_c = O.psnr(out["coarse"]["rgb"].cpu(), torch.from_numpy(g["coarse_rgb"]).reshape(-1, 3))
    st = robust_render_stats(out["fine"]["rgb"].cpu().numpy(), out["fine"]["depth"].cpu().numpy(), z_f.cpu().numpy(), g, span)
    print(f"ADV {name:24s} {prec:5s} {'folded' if fold else 'unfolded':8s} coarse PSNR {ps_c:6.1f} dB | fine PSNR {st['psnr']:6.1f} dB "
          f"(all rays {st['psnr_all']:6.1f}) depth p99/span {st['depth_p99_over_span']:.2e} bin-flip {st['bin_flip_frac']:.4f} "
          f"past-far rays {st['pastfar_frac']:.3f} (disagree {st['pastfar_disagree_frac']:.3f})")
    assert ps_c >= tol["psnr"], f"coarse PSNR {ps_c:.1f} dB"
    assert st["psnr"] >= tol["psnr"], st
    assert st["depth_p99_over_span"] <= tol["depth_p99"], st
    assert st["pastfar_frac"] <= 0.26, st  # only the forced rays (every 4th)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("name", ADVERSARIAL_SCENARIOS)
def test_surface_like_density_render_fp32_path(ops, dev, name, prec):
    """the exact path AND the shipped default ("f16x3": split operands + fp32 tables) at the same bars"""
    g, out, z_f, span = _render(ops, dev, name, prec, prec == "f16x3")
    st = robust_render_stats(out["fine"]["rgb"].cpu().numpy(), out["fine"]["depth"].cpu().numpy(), z_f.cpu().numpy(), g, span)
    print(f"ADV {name:24s} {prec} fine PSNR {st['psnr']:6.1f} dB (all rays {st['psnr_all']:6.1f}) depth p99/span "
          f"{st['depth_p99_over_span']:.2e} bin-flip {st['bin_flip_frac']:.4f} past-far rays {st['pastfar_frac']:.3f} "
          f"(disagree {st['pastfar_disagree_frac']:.3f})")
    assert st["psnr"] >= 70.0 and st["depth_p99_over_span"] <= 1e-3 and st["bin_flip_frac"] <= 0.01, st
