"""Shared test plumbing: load a golden fixture and rebuild the seeded scene / MLP weights it
was generated from (pixelnerf_amd.synthetic is deterministic, so only rays, noise and outputs
are stored in tests/golden/)."""
import functools
import os

import numpy as np
import torch

from testdata import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ADVERSARIAL_SCENARIOS = ["adv_surface_sn64", "adv_surface_srn", "adv_surface_dtu", "adv_surface_coarse_net"]

RENDER_SCENARIOS = [
    "sn64_c32", "sn64_64_128", "srn_mini_64_128", "dtu_mini_64_128", "train_64_32",
    "mv_mini_lindisp", "sn64_coarse_only_mlp",
    "dtu6_mini_64_128", "dtu9_mini_64_128",  # NS = 6 / 9: the reference's 6- and 9-view DTU evaluations (README.md:201-202)
]

# scenes with per-point stage fixtures (stages.npz + stages_manyview.npz); the multi-view ones also carry view-maximum outputs
STAGE_SCENES = ["sn64", "dtu_mini", "mv_mini", "dtu6_mini", "dtu9_mini"]
MV_STAGE_SCENES = ["dtu_mini", "mv_mini", "dtu6_mini", "dtu9_mini"]

# fixtures frozen in a second file so the first keeps regenerating bit-identically: loaded as one dict under the first's name
_MERGED = {"stages": ("stages_manyview",), "gradients": ("gradients_3view",)}


@functools.lru_cache(maxsize=None)
def _load_npz(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


def load_golden(name):
    g = dict(_load_npz(name))
    for extra in _MERGED.get(name, ()):
        e = _load_npz(extra)
        assert not set(e) & set(g), "fixture files share keys"
        g.update(e)
    if name == "combine_max":  # the many-view maxima live next to their points in stages_manyview.npz
        for k, v in _load_npz("stages_manyview").items():
            if "_max_" in k:
                g[k.replace("_max_", "_out_")] = v
    return g


@functools.lru_cache(maxsize=None)
def mlp_params(seed):
    return synthetic.make_mlp_params(int(seed))


@functools.lru_cache(maxsize=None)
def scene_for(name, seed=2):
    return synthetic.make_scene(name, seed=int(seed))


def golden_setup(name):
    """-> (g, scene, meta, mlp_coarse, mlp_fine|None, rays(SB,B,8), noise dict)"""
    g = load_golden(name)
    scene, meta = scene_for(str(g["scene"]), int(g["scene_seed"]))
    mc = mlp_params(int(g["mlp_seed_coarse"]))
    mf = mlp_params(int(g["mlp_seed_fine"])) if int(g["use_mlp_fine"]) else None
    if "sigma_gain" in g:  # adversarial fixtures: surface-like density variant of the seeded networks
        gain, tau = float(g["sigma_gain"]), float(g["sigma_tau"])
        mc = synthetic.surface_variant(mc, gain, tau)
        mf = None if mf is None else synthetic.surface_variant(mf, gain, tau)
    rays = torch.from_numpy(g["rays"])
    noise = {k[6:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("noise_")}
    return g, scene, meta, mc, mf, rays, noise


def assert_close_frac(actual, desired, atol, max_frac=0.0, loose_atol=None, what=""):
    """|actual-desired| <= atol everywhere except for at most `max_frac` of the elements,
    which must still be within `loose_atol`.  The fine pass places samples through a
    discontinuous map of the coarse weights (searchsorted bin index, nerf.py:138): a 1-ulp
    difference in a coarse weight can move one fine sample to the neighbouring bin, so
    end-to-end comparisons allow a tiny fraction of such flips (SURVEY.md §7 hard parts)."""
    a = np.asarray(actual, dtype=np.float64)
    d = np.asarray(desired, dtype=np.float64)
    assert a.shape == d.shape, (what, a.shape, d.shape)
    err = np.abs(a - d)
    assert np.isfinite(err).all(), f"{what}: non-finite values"
    bad = err > atol
    frac = bad.mean() if bad.size else 0.0
    assert frac <= max_frac, f"{what}: {bad.sum()}/{bad.size} elements exceed atol={atol} (max err {err.max():.3e})"
    if loose_atol is not None and bad.any():
        assert err.max() <= loose_atol, f"{what}: max err {err.max():.3e} > loose_atol={loose_atol}"


def robust_render_stats(rgb, depth, z_fine, g, span):
    """Fine-pass comparison that is meaningful on the adversarial fixtures.  A fine sample beyond `far`
    (searchsorted index == n_coarse, nerf.py:138-141) gives a NEGATIVE last delta (nerf.py:181) and, if the density
    there is positive, alpha = 1 - exp(+|delta| sigma) << 0: the ray's colour is then an ill-conditioned function of
    the inputs.  Whether such a sample exists hangs on `u >= cdf[-1]` with cdf[-1] = 1 +- 1 ulp, i.e. on the rounding
    of a 64-term sum -- two correct fp32 implementations (the reference on CPU vs on GPU, or vs the oracle) disagree on
    it for a few rays (`pastfar_disagree_frac`), and where both have the sample, a density that is 0 on one side and
    positive on the other changes the colour by orders of magnitude.  Rays on which EITHER side has a beyond-far
    sample are therefore counted (`pastfar_frac`; only the forced rays, every 4th, can be among them) and excluded
    from the PSNR / depth figures (`psnr_all` keeps them, for the record).
    -> dict(psnr, psnr_all, depth_p99_over_span, pastfar_frac, pastfar_disagree_frac, bin_flip_frac, n_rays)"""
    rgb = np.asarray(rgb, np.float64).reshape(-1, 3)
    depth = np.asarray(depth, np.float64).reshape(-1)
    z = np.asarray(z_fine, np.float64).reshape(rgb.shape[0], -1)
    zg = g["fine_z"].astype(np.float64).reshape(z.shape)
    far = g["rays"].reshape(-1, 8)[:, 7].astype(np.float64)
    dis = (z[:, -1] > far) != (zg[:, -1] > far)
    ok = ~((z[:, -1] > far) | (zg[:, -1] > far))
    mse = lambda a, b: float(np.mean((a - b) ** 2))  # noqa: E731
    ref_rgb, ref_d = g["fine_rgb"].reshape(-1, 3).astype(np.float64), g["fine_depth"].reshape(-1).astype(np.float64)
    return dict(
        psnr=-10.0 * np.log10(max(mse(rgb[ok], ref_rgb[ok]), 1e-30)),
        psnr_all=-10.0 * np.log10(max(mse(rgb, ref_rgb), 1e-30)),
        depth_p99_over_span=float(np.percentile(np.abs(depth[ok] - ref_d[ok]), 99)) / span,
        pastfar_frac=float(1.0 - ok.mean()), pastfar_disagree_frac=float(dis.mean()),
        bin_flip_frac=float((np.abs(z - zg) > 1e-4 * span).mean()),
        n_rays=int(rgb.shape[0]))


# gradient-only scenarios of tests/golden/gradients.npz (oracle/make_goldens.py GRAD_ONLY): no render fixture -- rays, noise
# and networks are regenerated from their seeds exactly as the generator does
GRAD_ONLY = {"train_cfg5": ("train", 64, 32, 16, 128, False, True),   # BASELINE configs[4] at full size: 4 objects x 128 rays
             "train_mv3": ("train_mv3", 64, 32, 16, 64, False, True)}  # DTU-style step: 2 objects x 3 source views x 64 rays
# every scenario with gradients frozen from the reference's own autograd (gradients.npz + gradients_3view.npz)
GRAD_SCENARIOS = ["train_64_32", "srn_mini_64_128", "train_cfg5", "dtu_mini_64_128", "train_mv3"]


def grad_setup(name):
    """-> (g-like dict, scene, meta, mlp_coarse, mlp_fine, rays (SB,B,8), noise) for a render golden OR a gradient-only scenario"""
    if name not in GRAD_ONLY:
        return golden_setup(name)
    scene_name, Kc, Kf, Kfd, n_rays, lindisp, _ = GRAD_ONLY[name]
    scene, meta = scene_for(scene_name, 2)
    rays = synthetic.target_rays(meta, n_rays=n_rays)
    noise = synthetic.make_noise(rays.shape[0] * n_rays, Kc, Kf, Kfd)
    g = dict(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, white_bkgd=meta["white_bkgd"], lindisp=lindisp)
    return g, scene, meta, mlp_params(11), mlp_params(12), rays, noise
