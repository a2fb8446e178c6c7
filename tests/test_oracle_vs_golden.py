"""
CPU tests: the oracle restatement (oracle/pnr_oracle.py) against the golden vectors frozen
from the UNMODIFIED reference (oracle/make_goldens.py -> tests/golden/*.npz).

Tolerances: both sides are torch CPU fp32 running the same arithmetic in (almost) the same
order, so agreement is at rounding level: 2e-5 abs on rgb/sigma/weights (values in [0,1]),
1e-5 * (far-near) on z.  The explicit bilinear lookup vs F.grid_sample differs only in
fp32 rounding of the corner weights.
"""
import numpy as np
import pytest
import torch

from helpers import (GRAD_SCENARIOS, MV_STAGE_SCENES, STAGE_SCENES, ADVERSARIAL_SCENARIOS, RENDER_SCENARIOS, assert_close_frac, robust_render_stats, golden_setup, load_golden, mlp_params,
                     scene_for)
from oracle import pnr_oracle as O
from testdata import synthetic


def test_positional_encoding_matches_reference():
    g = load_golden("stages")
    out = O.positional_encoding(torch.from_numpy(g["posenc_x"]))
    assert out.shape == (257, 39)
    np.testing.assert_allclose(out.numpy(), g["posenc_out"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("scene_name", STAGE_SCENES)
def test_index_latent_matches_reference_grid_sample(scene_name):
    g = load_golden("stages")
    scene, _ = scene_for(scene_name)
    out = O.index_latent(scene["latent"], torch.from_numpy(g[f"{scene_name}_uv"]),
                         scene["image_shape"])
    np.testing.assert_allclose(out.numpy(), g[f"{scene_name}_index"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("scene_name", STAGE_SCENES)
def test_pixelnerf_forward_matches_reference(scene_name):
    g = load_golden("stages")
    scene, _ = scene_for(scene_name)
    xyz = torch.from_numpy(g[f"{scene_name}_xyz"])
    vd = torch.from_numpy(g[f"{scene_name}_viewdirs"])
    for which, seed in (("coarse", 11), ("fine", 12)):
        out = O.pixelnerf_forward(scene, mlp_params(seed), xyz, vd)
        ref = g[f"{scene_name}_out_{which}"]
        np.testing.assert_allclose(out[..., :3].numpy(), ref[..., :3], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out[..., 3].numpy(), ref[..., 3], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("scene_name", MV_STAGE_SCENES)
def test_pixelnerf_forward_with_max_pooling_matches_reference(scene_name):
    """combine_type = "max" (src/util/util.py:467-468): the reference's own outputs with both ResnetFCs switched to the view
    maximum (tests/golden/combine_max.npz) -- and they differ from the view mean by O(1), so the branch is really exercised"""
    g, gm = load_golden("stages"), load_golden("combine_max")
    scene, _ = scene_for(scene_name)
    xyz = torch.from_numpy(g[f"{scene_name}_xyz"])
    vd = torch.from_numpy(g[f"{scene_name}_viewdirs"])
    for which, seed in (("coarse", 11), ("fine", 12)):
        out = O.pixelnerf_forward(scene, mlp_params(seed), xyz, vd, combine_type="max")
        ref = gm[f"{scene_name}_out_{which}"]
        np.testing.assert_allclose(out.numpy()[..., :3], ref[..., :3], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out.numpy()[..., 3], ref[..., 3], rtol=1e-5, atol=2e-5)
        assert np.abs(ref - g[f"{scene_name}_out_{which}"]).max() > 0.5


@pytest.mark.parametrize("name", sorted(synthetic.VARIANTS))
def test_pixelnerf_forward_of_other_model_confs_matches_reference(name):
    """model confs outside the shipped one (models.py:22-65: coded view directions -- the reference's default --, camera-space
    positions, depth-only feature, Softplus / SPADE / max pooling, a global latent, no encoder; other ResnetFC shapes): the
    reference's own outputs (tests/golden/variants.npz) against the oracle's general restatement"""
    g = load_golden("variants")
    scene, _, xyz, vd, glob = synthetic.variant_inputs(name)
    conf = synthetic.variant_model_conf(name)
    params = synthetic.variant_mlp_params(name, int(g[f"{name}_d_in"]), int(g[f"{name}_d_latent"]))
    for which, p in zip(("coarse", "fine"), params):
        out = O.pixelnerf_forward_general(scene, p, xyz, vd, conf, global_latent=glob).numpy()
        ref = g[f"{name}_out_{which}"]
        assert out.shape == ref.shape
        np.testing.assert_allclose(out[..., :3], ref[..., :3], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out[..., 3], ref[..., 3], rtol=1e-4, atol=2e-4)
    assert np.abs(g[f"{name}_out_coarse"] - g[f"{name}_out_fine"]).max() > 1e-3  # two different networks


def test_points_on_and_behind_a_camera_plane_match_reference():
    """models.py:206-212 has no frustum culling: camera-space z == 0 gives u = x/0 = +-inf (border clamp) or 0/0 = NaN,
    which ATen's grid_sample maps to coordinate 0; z > 0 mirrors.  The reference's outputs on such points are finite and
    the oracle reproduces them (exactly on the degenerate rows)."""
    g = load_golden("adv_plane")
    scene, _ = scene_for("plane_mini")
    xyz, vd = torch.from_numpy(g["xyz"]), torch.from_numpy(g["viewdirs"])
    assert np.isfinite(g["out_coarse"]).all() and np.isfinite(g["out_fine"]).all()
    for which, seed in (("coarse", 11), ("fine", 12)):
        out = O.pixelnerf_forward(scene, mlp_params(seed), xyz, vd)
        np.testing.assert_allclose(out[..., :3].numpy(), g[f"out_{which}"][..., :3], rtol=0, atol=2e-5)
        np.testing.assert_allclose(out[..., 3].numpy(), g[f"out_{which}"][..., 3], rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("name", RENDER_SCENARIOS + ADVERSARIAL_SCENARIOS)
def test_sampling_and_compositing_stagewise_match_reference(name):
    """Stage-wise, fed with the reference's own intermediates so that no discontinuity can
    amplify rounding: coarse z from u1; fine z from the golden coarse weights/depth;
    compositing from the golden per-point rgb/sigma."""
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    lindisp, white = bool(g["lindisp"]), bool(g["white_bkgd"])
    span = float(meta["z_far"] - meta["z_near"])
    r = rays.reshape(-1, 8)
    zc = O.sample_coarse(r, noise["u1"], Kc, lindisp)
    np.testing.assert_allclose(zc.numpy(), g["coarse_z"], rtol=0, atol=1e-6 * span)
    w, rgb, depth = O.composite_from_rgbsigma(r, torch.from_numpy(g["coarse_z"]),
                                              torch.from_numpy(g["coarse_rgbsigma"]), white)
    np.testing.assert_allclose(w.numpy(), g["coarse_weights"].reshape(-1, Kc), rtol=0, atol=1e-6)
    np.testing.assert_allclose(rgb.numpy(), g["coarse_rgb"].reshape(-1, 3), rtol=0, atol=2e-6)
    np.testing.assert_allclose(depth.numpy(), g["coarse_depth"].reshape(-1), rtol=0, atol=1e-5)
    if Kf > 0:
        samps = [torch.from_numpy(g["coarse_z"])]
        wc = torch.from_numpy(g["coarse_weights"]).reshape(-1, Kc)
        if Kf - Kfd > 0:
            samps.append(O.sample_fine(r, wc, noise["u2"], noise["u3"], Kc, lindisp))
        if Kfd > 0:
            samps.append(O.sample_fine_depth(r, torch.from_numpy(g["coarse_depth"]).reshape(-1),
                                             noise["n4"], float(g["depth_std"])))
        zf = torch.sort(torch.cat(samps, -1), dim=-1)[0]
        np.testing.assert_allclose(zf.numpy(), g["fine_z"], rtol=0, atol=1e-6 * span)
        w, rgb, depth = O.composite_from_rgbsigma(r, torch.from_numpy(g["fine_z"]),
                                                  torch.from_numpy(g["fine_rgbsigma"]), white)
        np.testing.assert_allclose(w.numpy(), g["fine_weights"].reshape(-1, Kc + Kf), rtol=0,
                                   atol=1e-6)
        np.testing.assert_allclose(rgb.numpy(), g["fine_rgb"].reshape(-1, 3), rtol=0, atol=2e-6)


@pytest.mark.parametrize("name", RENDER_SCENARIOS + ADVERSARIAL_SCENARIOS)
def test_render_end_to_end_matches_reference(name):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    out = O.render(scene, mc, mf, rays, noise, Kc, Kf, Kfd, depth_std=float(g["depth_std"]),
                   white_bkgd=bool(g["white_bkgd"]), lindisp=bool(g["lindisp"]))
    span = float(meta["z_far"] - meta["z_near"])
    SB = rays.shape[0]
    assert ("fine" in out) == (Kf > 0)
    # coarse pass: continuous in its inputs -> rounding-level agreement everywhere
    K = Kc
    np.testing.assert_allclose(out["coarse"]["z"].reshape(-1, K).numpy(), g["coarse_z"], rtol=0,
                               atol=1e-6 * span)
    np.testing.assert_allclose(out["coarse"]["rgbsigma"].reshape(-1, K, 4)[..., :3].numpy(),
                               g["coarse_rgbsigma"][..., :3], rtol=0, atol=3e-5)
    np.testing.assert_allclose(out["coarse"]["weights"].numpy(), g["coarse_weights"], rtol=0,
                               atol=5e-5)
    np.testing.assert_allclose(out["coarse"]["rgb"].numpy(), g["coarse_rgb"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(out["coarse"]["depth"].numpy(), g["coarse_depth"], rtol=0,
                               atol=5e-5 * span)
    assert out["coarse"]["rgb"].shape == (SB, rays.shape[1], 3)
    if Kf > 0 and name in ADVERSARIAL_SCENARIOS:
        # peaked density + a draw at the top of the cdf: see helpers.robust_render_stats
        st = robust_render_stats(out["fine"]["rgb"].numpy(), out["fine"]["depth"].numpy(), out["fine"]["z"].numpy(), g, span)
        assert st["pastfar_frac"] <= 0.26 and st["bin_flip_frac"] <= 0.01, st  # only the forced rays (every 4th)
        assert st["psnr"] >= 70.0 and st["depth_p99_over_span"] <= 1e-3, st
    elif Kf > 0:
        # fine pass: allow <=0.2% of samples to sit in a neighbouring importance bin
        K = Kc + Kf
        assert_close_frac(out["fine"]["z"].reshape(-1, K).numpy(), g["fine_z"], 1e-5 * span,
                          max_frac=2e-3, loose_atol=span / Kc * 1.01, what="fine z")
        assert_close_frac(out["fine"]["rgb"].numpy(), g["fine_rgb"], 5e-5, max_frac=2e-2,
                          loose_atol=2e-2, what="fine rgb")
        assert_close_frac(out["fine"]["depth"].numpy(), g["fine_depth"], 5e-5 * span,
                          max_frac=2e-2, loose_atol=2e-2 * span, what="fine depth")
        assert O.psnr(out["fine"]["rgb"], torch.from_numpy(g["fine_rgb"])) > 70.0


# ------------------------------------------------------------------ neighbours of the path


@pytest.mark.parametrize("name", ["pool", "nopool"])
def test_encoder_format_matches_reference(name):
    """Restated encoder output formatting vs SpatialEncoder.forward of the reference run on the same
    seeded stage tensors (oracle/make_goldens.py::neighbour_goldens)."""
    from testdata import synthetic
    g = load_golden("neighbours")
    lat, scaling = O.encoder_format(synthetic.pyramid_stages(name))
    np.testing.assert_array_equal(lat.numpy(), g[f"pyr_{name}_latent"])
    np.testing.assert_array_equal(scaling.numpy(), g[f"pyr_{name}_scaling"])


def test_gen_rays_restatement_matches_reference():
    from testdata import synthetic
    g = load_golden("neighbours")
    rays = synthetic.gen_rays(torch.from_numpy(g["rays_poses"]), 20, 15, torch.from_numpy(g["rays_focal"]), 0.8, 1.8,
                              c=torch.from_numpy(g["rays_c"]))
    np.testing.assert_allclose(rays.numpy(), g["rays_out"], rtol=0, atol=1e-6)


def test_eval_epilogue_psnr_matches_reference_util_psnr():
    """util.psnr (fp32 mean) vs the skimage-style fp64 restatement: same value to ~1e-5 dB."""
    g = load_golden("neighbours")
    out = O.eval_epilogue(g["psnr_pred"], np.zeros((3, 300), np.float32), 0.0, 1.0, gt=g["psnr_gt"])
    np.testing.assert_allclose(out["psnr"], g["psnr_out"], rtol=0, atol=1e-4)
    assert out["rgb_u8"].dtype == np.uint8 and out["rgb"].min() >= 0.0 and out["rgb"].max() <= 1.0


def test_bbox_pixels_matches_reference_bbox_sample():
    g = load_golden("neighbours")
    pix = O.bbox_pixels(torch.from_numpy(g["bbox_boxes"]), torch.from_numpy(g["bbox_ids"]), torch.from_numpy(g["bbox_ux"]),
                        torch.from_numpy(g["bbox_uy"]))
    np.testing.assert_array_equal(pix.numpy(), g["bbox_pix"])


# ------------------------------------------------------------------ gradients (BASELINE config 5)


@pytest.mark.parametrize("name", GRAD_SCENARIOS)  # incl. the 3-view scenarios of gradients_3view.npz (README.md:204)
def test_oracle_autograd_matches_reference_autograd(name):
    """torch autograd through the oracle vs the UNMODIFIED reference's own backward (tests/golden/gradients.npz:
    per-tensor L2 norm + seeded subsample of every ResnetFC gradient of both networks and of encoder.latent,
    including the position gradient through the depth samples, nerf.py:292).  fp32 on both sides, different
    summation orders: norms within 1e-4, subsamples within 1e-3 relative (measured 1e-6 / 2e-4)."""
    from testdata import synthetic
    from helpers import grad_setup
    gg = load_golden("gradients")
    g, scene, meta, mc, mf, rays, noise = grad_setup(name)  # train_cfg5: BASELINE configs[4] at full size (4 x 128 rays)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    sc = dict(scene)
    sc["latent"] = scene["latent"].clone().requires_grad_(True)
    pc = {k: v.clone().requires_grad_(True) for k, v in mc.items()}
    pf = {k: v.clone().requires_grad_(True) for k, v in mf.items()}
    out = O.render(sc, pc, pf, rays, noise, Kc, Kf, Kfd, white_bkgd=bool(g["white_bkgd"]), lindisp=bool(g["lindisp"]))
    gt = torch.from_numpy(gg[f"{name}_gt"])
    loss = ((out["coarse"]["rgb"] - gt) ** 2).mean() + ((out["fine"]["rgb"] - gt) ** 2).mean()
    loss.backward()
    assert abs(loss.item() - float(gg[f"{name}_loss"])) <= 1e-6
    grads = {"latent": sc["latent"].grad, **{"coarse." + k: v.grad for k, v in pc.items()},
             **{"fine." + k: v.grad for k, v in pf.items()}}
    assert len(grads) == 61
    for key, gr in grads.items():
        flat = gr.reshape(-1).numpy()
        ref_s = gg[f"{name}_grad_{key}_sample"]
        ref_n = float(gg[f"{name}_grad_{key}_norm"])
        got_s = flat[synthetic.grad_sample_index(flat.size, key)]
        assert ref_n > 0, key
        assert abs(np.linalg.norm(flat.astype(np.float64)) - ref_n) <= 1e-4 * ref_n, key
        assert np.linalg.norm(got_s - ref_s) <= 1e-3 * np.linalg.norm(ref_s), key
