"""Shared test plumbing: load a golden fixture and rebuild the seeded scene / MLP weights it
was generated from (pixelnerf_amd.synthetic is deterministic, so only rays, noise and outputs
are stored in tests/golden/)."""
import functools
import os

import numpy as np
import torch

from testdata import synthetic

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

RENDER_SCENARIOS = [
    "sn64_c32", "sn64_64_128", "srn_mini_64_128", "dtu_mini_64_128", "train_64_32",
    "mv_mini_lindisp", "sn64_coarse_only_mlp",
]


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return {k: z[k] for k in z.files}


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
