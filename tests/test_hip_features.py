"""
GPU tests (-m gpu) of the feature phase IN ISOLATION (SURVEY.md 8a rows R7 + R8), against goldens frozen from the
reference's own modules (tests/golden/stages.npz, oracle/make_goldens.py):
  * PositionalEncoding.forward (src/model/code.py:30-42): `posenc_x` -> `posenc_out`, through pnr_point_features_f32
    (the exact-fp32 path's feature kernel, libm sinf) with an identity source camera, and through the fused 16-bit
    kernel's feature phase (hardware sine, code rounded to f16 on its way to LDS) read back from the training dump d_in;
  * SpatialEncoder.index (src/model/encoder.py:80-109): `<scene>_uv` -> `<scene>_index`, through the same entry with points
    constructed to project onto those pixel coordinates.
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, mlp_params, scene_for, STAGE_SCENES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from pixelnerf_amd import ops as _ops
    return _ops


def identity_scene(ops, dev, name="sn64"):
    """the named scene's grid seen by ONE source camera at the origin looking down -z (world == camera frame)"""
    s, meta = scene_for(name)
    pose = torch.eye(4)[:3].reshape(1, 3, 4).contiguous()
    return s, ops.make_scene(s["latent"][:1].to(dev), pose.to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], 1)


def test_positional_encoding_fp32_matches_reference(ops, dev):
    g = load_golden("stages")
    _, sc = identity_scene(ops, dev)
    x = torch.from_numpy(g["posenc_x"]).to(dev)[None]  # (1, 257, 3)
    d = torch.nn.functional.normalize(torch.randn(1, x.shape[1], 3, generator=torch.Generator().manual_seed(0)), dim=-1).to(dev)
    in42, _ = ops.point_features(sc, x, d)
    got = in42[0, 0].cpu().numpy()
    # code.py:37-41: [x, sin(f_k x), sin(f_k x + pi/2)]; device libm sinf vs the reference's CPU sin: a few ulp
    np.testing.assert_allclose(got[:, :39], g["posenc_out"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(got[:, 39:42], d[0].cpu().numpy(), rtol=0, atol=1e-7)  # identity rotation of the view direction
    assert (got[:, 42:] == 0).all()


def test_positional_encoding_of_the_fused_kernel_matches_reference(ops, dev):
    """the 16-bit kernels' feature phase (v_sin_f32, values rounded to f16): read from the training dump of lin_in's operand"""
    g = load_golden("stages")
    _, sc = identity_scene(ops, dev)
    x = torch.from_numpy(g["posenc_x"]).to(dev)
    R = x.shape[0]
    d = torch.nn.functional.normalize(torch.randn(R, 3, generator=torch.Generator().manual_seed(0)), dim=-1).to(dev)
    rays = torch.cat([x, d, torch.zeros(R, 2, device=dev)], dim=1).contiguous()  # origin = point, z = 0 (autograd._PointsFunction)
    z = torch.zeros(R, 1, device=dev)
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    _, dumps = ops.eval_ray_samples_train(sc, ops.pack_mlp(state, "f16"), rays, z)
    got = dumps.d_in.float().cpu().numpy()
    ref = g["posenc_out"]
    # f16 rounding of values in [-1,1] U the identity part: half an f16 ulp of max(|x|, 1) + the hardware sine's 1e-5
    tol = 2.0 ** -11 * np.maximum(1.0, np.abs(ref)) + 2e-5
    assert (np.abs(got[:, :39] - ref) <= tol).all(), float(np.abs(got[:, :39] - ref).max())
    dumps.release()


def _posenc_restated(x, freqs2, phases2, include_input):
    """src/model/code.py:30-42 in torch ops (test-side restatement; runs on whatever device x is on)"""
    if freqs2.numel() == 0:
        return x if include_input else x.new_zeros(x.shape[0], 0)
    e = x.unsqueeze(1).repeat(1, freqs2.numel(), 1)
    e = torch.sin(torch.addcmul(phases2.reshape(1, -1, 1), e, freqs2.reshape(1, -1, 1))).reshape(x.shape[0], -1)
    return torch.cat((x, e), dim=-1) if include_input else e


def test_positional_encoding_module_runs_the_hip_operator(ops, dev):
    """PositionalEncoding.forward on its own = pnr_positional_encoding, against the reference module's frozen output"""
    from pixelnerf_amd.model.code import PositionalEncoding
    g = load_golden("stages")
    code = PositionalEncoding(num_freqs=6, d_in=3, freq_factor=1.5, include_input=True).to(dev)
    x = torch.from_numpy(g["posenc_x"]).to(dev)
    out = code(x)
    assert out.shape == (x.shape[0], 39) and out.is_cuda
    np.testing.assert_allclose(out.cpu().numpy(), g["posenc_out"], rtol=0, atol=1e-6)
    assert torch.equal(out[:, :3], x)
    assert code(x[:0]).shape == (0, 39)  # empty batch
    with pytest.raises(ValueError):
        code(x[:, :2])


@pytest.mark.parametrize("d_in,F,include", [(3, 6, True), (2, 4, False), (5, 1, True), (3, 0, True)])
def test_positional_encoding_operator_forward_backward(ops, dev, d_in, F, include):
    from pixelnerf_amd.model.code import PositionalEncoding
    code = PositionalEncoding(num_freqs=F, d_in=d_in, freq_factor=1.5, include_input=include).to(dev)
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(1000, d_in, generator=gen) * 4 - 2).to(dev).requires_grad_(True)
    gw = torch.randn(1000, code.d_out, generator=gen).to(dev)
    out = code(x)
    (out * gw).sum().backward()
    # the reference lines on the CPU (ATen's CPU addcmul is one fused multiply-add, like the goldens and like the kernel; the
    # eager GPU addcmul rounds the product first and lands up to an ulp of the ARGUMENT away: 8e-6 at |x f| ~ 100)
    x2 = x.detach().cpu().requires_grad_(True)
    ref = _posenc_restated(x2, code._freqs.cpu(), code._phases.cpu(), include)
    (ref * gw.cpu()).sum().backward()
    assert out.shape == ref.shape
    if out.numel():
        assert (out.detach().cpu() - ref.detach()).abs().max().item() <= 1e-6
    scale = max(1.0, float(x2.grad.abs().max()))
    assert (x.grad.cpu() - x2.grad).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("name", STAGE_SCENES)
def test_spatial_encoder_index_matches_reference(ops, dev, name):
    g = load_golden("stages")
    s, meta = scene_for(name)
    uv = torch.from_numpy(g[name + "_uv"])        # (NV, 64, 2) pixel coordinates, some outside the image (border clamp)
    ref = torch.from_numpy(g[name + "_index"])    # (NV, 512, 64)
    fx, fy = float(s["focal"][0, 0]), float(-s["focal"][0, 1])
    cx, cy = float(s["c"][0, 0]), float(s["c"][0, 1])
    pose = torch.eye(4)[:3].reshape(1, 3, 4).contiguous()
    for v in range(uv.shape[0]):
        sc = ops.make_scene(s["latent"][v:v + 1].to(dev), pose.to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], 1)
        # camera at the origin, depth -1: u = x fx + cx, v = -y fy + cy  (models.py:206-212 with focal stored as (fx, -fy))
        x = (uv[v, :, 0] - cx) / fx
        y = -(uv[v, :, 1] - cy) / fy
        pts = torch.stack([x, y, -torch.ones_like(x)], dim=-1)[None].contiguous().to(dev)
        _, zlat = ops.point_features(sc, pts, torch.zeros_like(pts))
        got = zlat[0, 0].cpu().t()  # (512, 64)
        # reconstructing uv from (x, y) costs ~1e-5 px; the grid is O(1) per texel step
        assert (got - ref[v]).abs().max() <= 1e-4, float((got - ref[v]).abs().max())


def _encoder_with_grid(dev, s):
    import warnings
    from pixelnerf_amd.model.encoder import SpatialEncoder
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        enc = SpatialEncoder("resnet34", pretrained=False, use_first_pool=False).to(dev).eval()
    lat = s["latent"].to(dev)
    enc.latent = lat
    wh = torch.tensor([float(lat.shape[-1]), float(lat.shape[-2])], device=dev)
    enc.latent_scaling = wh / (wh - 1) * 2.0
    return enc


@pytest.mark.parametrize("name", STAGE_SCENES)
def test_spatial_encoder_index_operator_matches_reference(dev, name):
    """SpatialEncoder.index called on its own (src/model/encoder.py:80-109) is the HIP operator pnr_grid_index: against the
    outputs of the reference's own `index` on the same grid and pixel coordinates (goldens), incl. points outside the image."""
    g = load_golden("stages")
    s, meta = scene_for(name)
    enc = _encoder_with_grid(dev, s)
    uv = torch.from_numpy(g[name + "_uv"]).to(dev)
    ref = torch.from_numpy(g[name + "_index"])
    with torch.no_grad():
        got = enc.index(uv, None, s["image_shape"].to(dev))
    assert got.shape == ref.shape
    assert (got.cpu() - ref).abs().max() <= 2e-6 * max(1.0, float(ref.abs().max())), float((got.cpu() - ref).abs().max())
    # one uv set for all views (encoder.py:91-92) and already-normalised coordinates (no image_size)
    with torch.no_grad():
        one = enc.index(uv[:1], None, s["image_shape"].to(dev))
        nrm = enc.index(uv * (enc.latent_scaling / s["image_shape"].to(dev)) - 1.0)
    assert torch.equal(one[0], got[0]) and torch.equal(nrm, got)


@pytest.mark.parametrize("NV,C,H,W,N", [(2, 128, 9, 13, 200), (3, 70, 5, 7, 130), (1, 512, 2, 2, 1)])  # ragged channel / point counts too
def test_spatial_encoder_index_operator_gradients(dev, NV, C, H, W, N):
    """gradients of the stand-alone lookup to the grid and to the coordinates against torch autograd through F.grid_sample
    (what the reference's index() differentiates through), interior and clamped points"""
    import torch.nn.functional as F
    gen = torch.Generator().manual_seed(2)
    lat = torch.randn(NV, C, H, W, generator=gen)
    uv = torch.rand(NV, N, 2, generator=gen) * 2.6 - 1.3  # a fifth of the points beyond the border
    gw = torch.randn(NV, C, N, generator=gen)
    a, b = lat.clone().requires_grad_(True), uv.clone().requires_grad_(True)
    ref = F.grid_sample(a, b.unsqueeze(2), align_corners=True, mode="bilinear", padding_mode="border")[..., 0]
    (ref * gw).sum().backward()
    s = {"latent": lat}
    enc = _encoder_with_grid(dev, s)
    la, ub = lat.to(dev).requires_grad_(True), uv.to(dev).requires_grad_(True)
    enc.latent = la
    got = enc.index(ub)
    (got * gw.to(dev)).sum().backward()
    assert (got.detach().cpu() - ref.detach()).abs().max() <= 2e-6 * float(ref.detach().abs().max())
    assert (la.grad.cpu() - a.grad).abs().max() <= 1e-5 * max(1.0, float(a.grad.abs().max()))
    assert (ub.grad.cpu() - b.grad).abs().max() <= 2e-5 * max(1.0, float(b.grad.abs().max()))
