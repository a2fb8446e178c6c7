"""
GPU parity (-m gpu): a seeded random sweep of the fp32 renderer stages (stratified sampling, importance +
depth sampling with the sort, alpha compositing) over ragged shapes and hostile value ranges, against the
CPU oracle.  Complements the fixed golden scenarios with sizes/values they do not contain: K = 1, non-power-
of-two K up to the sampler limits, near == far rays, huge / zero / negative densities, weights that are all
zero, uniforms at 0 and 1-ulp.

Tolerances (fp32 on both sides): z 2e-6*far (+ 4 ulp(t) * far^2 (1/near - 1/far) for lindisp, whose map
t -> z is ill-conditioned near far: with near 0.5 / far 50 one ulp of t moves z by 3e-4); weights 2e-6; rgb 1e-5;
depth 2e-5*far;
importance samples may land in the neighbouring cdf bin for <= 0.5 % of the draws (cdf rounding ties).
"""
import numpy as np
import pytest
import torch

from helpers import assert_close_frac
from oracle import pnr_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from pixelnerf_amd import ops as _ops
    return _ops


def rand_rays(rs, R, near, far):
    o = rs.uniform(-2, 2, (R, 3))
    d = rs.randn(R, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    nf = np.stack([np.full(R, near), np.full(R, far)], 1)
    return torch.from_numpy(np.concatenate([o, d, nf], 1).astype(np.float32))


CASES = [  # (R, Kc, Kimp, Kfd, lindisp, white, near, far)
    (1, 1, 0, 0, False, True, 1.2, 4.0),
    (7, 3, 5, 2, False, False, 0.1, 5.0),
    (33, 17, 0, 9, True, True, 0.8, 1.8),
    (64, 64, 112, 16, False, True, 1.2, 4.0),
    (129, 31, 64, 0, True, False, 0.5, 50.0),
    (5, 256, 240, 16, False, True, 1.2, 4.0),   # round 1-3's sampler limits: n_coarse 256, total 512
    (3, 400, 600, 24, False, True, 1.2, 4.0),   # beyond them: 1024 samples per ray (dynamic LDS)
    (2, 1024, 2000, 48, True, False, 0.5, 6.0),  # 3072 samples per ray, 64 KiB of LDS per workgroup
    (300, 8, 1, 1, False, False, 2.0, 2.0),     # near == far: every z identical, zero-length intervals
]


@pytest.mark.parametrize("case", CASES, ids=[f"R{c[0]}_Kc{c[1]}_Ki{c[2]}_Kd{c[3]}{'_lin' if c[4] else ''}" for c in CASES])
def test_stage_sweep(ops, dev, case):
    R, Kc, Kimp, Kfd, lindisp, white, near, far = case
    rs = np.random.RandomState(R * 1000 + Kc)
    rays = rand_rays(rs, R, near, far)
    span = max(far - near, 1e-3)
    ztol = 2e-6 * max(span, far)
    if lindisp:  # z = 1/a(t) amplifies a 1-ulp difference in t by dz/dt = z^2 (1/near - 1/far): allow 4 ulp(t) at z = far
        ztol += 4 * 6e-8 * far * far * (1.0 / near - 1.0 / far)
    u1 = torch.from_numpy(rs.uniform(0, 1, (R, Kc)).astype(np.float32))
    u1[0, 0] = 0.0
    u1[-1, -1] = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    z_c = ops.sample_coarse(rays.to(dev), u1.to(dev), lindisp).cpu()
    z_ref = O.sample_coarse(rays, u1, Kc, lindisp)
    np.testing.assert_allclose(z_c.numpy(), z_ref.numpy(), rtol=0, atol=ztol)

    # compositing on hostile rgb/sigma: huge, zero and negative densities, colours outside [0,1]
    rgbs = torch.from_numpy(rs.uniform(-0.5, 1.5, (R, Kc, 4)).astype(np.float32))
    sig = rs.lognormal(0.0, 3.0, (R, Kc)).astype(np.float32)
    sig[rs.uniform(size=sig.shape) < 0.3] = 0.0
    sig[rs.uniform(size=sig.shape) < 0.1] *= -1.0
    if R > 2:
        sig[1] = 0.0          # a fully transparent ray: all weights 0 -> uniform pdf
        sig[2] = 1e6          # an opaque ray: first sample takes everything
    rgbs[..., 3] = torch.from_numpy(sig)
    w, rgb, depth = ops.composite(rays.to(dev), z_ref.to(dev), rgbs.to(dev), white_bkgd=white)
    w_ref, rgb_ref, depth_ref = O.composite_from_rgbsigma(rays, z_ref, rgbs, white)
    np.testing.assert_allclose(w.cpu().numpy(), w_ref.numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(rgb.cpu().numpy(), rgb_ref.numpy(), rtol=0, atol=1e-5)
    np.testing.assert_allclose(depth.cpu().numpy(), depth_ref.numpy(), rtol=0, atol=2e-5 * far)

    if Kimp + Kfd == 0:
        return
    u2 = torch.from_numpy(rs.uniform(0, 1, (R, Kimp)).astype(np.float32)) if Kimp else None
    u3 = torch.from_numpy(rs.uniform(0, 1, (R, Kimp)).astype(np.float32)) if Kimp else None
    n4 = torch.from_numpy(rs.randn(R, Kfd).astype(np.float32)) if Kfd else None
    if Kimp:
        u2[0, 0] = 0.0
        u2[-1, -1] = float(np.nextafter(np.float32(1.0), np.float32(0.0)))
    z_f = ops.sample_fine(rays.to(dev), w_ref.to(dev), depth_ref.to(dev), z_ref.to(dev),
                          None if u2 is None else u2.to(dev), None if u3 is None else u3.to(dev),
                          None if n4 is None else n4.to(dev), depth_std=0.01, lindisp=lindisp, want_ranks=bool(Kfd))
    ranks = None
    if Kfd:
        z_f, ranks = z_f
    parts = [z_ref]
    if Kimp:
        parts.append(O.sample_fine(rays, w_ref, u2, u3, Kc, lindisp))
    if Kfd:
        zd = O.sample_fine_depth(rays, depth_ref, n4, 0.01)
        parts.append(zd)
    z_all_ref, _ = torch.sort(torch.cat(parts, -1), dim=-1)
    z_f = z_f.cpu()
    assert z_f.shape == z_all_ref.shape
    assert (np.diff(z_f.numpy(), axis=1) >= 0).all()
    assert_close_frac(z_f.numpy(), z_all_ref.numpy(), ztol, max_frac=5e-3, loose_atol=max(span, far) / Kc * 1.01 + ztol,
                      what="fine z")
    if ranks is not None:  # the recorded positions really hold the depth samples
        got = torch.gather(z_f, 1, ranks.cpu().long())
        np.testing.assert_allclose(np.sort(got.numpy(), 1), np.sort(zd.numpy(), 1), rtol=0, atol=ztol)
