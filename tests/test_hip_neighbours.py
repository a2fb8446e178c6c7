"""
GPU parity tests (-m gpu) of the rows next to the hot path: whole-view rendering from
camera poses (rank 1), encoder output formatting (rank 2), eval epilogue (rank 3).  Checked against
fixtures frozen from the reference's own code (tests/golden/neighbours.npz) and the oracle restatements.

Tolerances: all fp32 on both sides.  Ray generation 1e-6; the bilinear upsample follows ATen's operation
order without FMA contraction: <= 2e-6 abs on N(0,1) data (stage 0 is copied exactly); clamp / uint8 /
depth normalisation exact; PSNR (fp64 accumulation in a fixed order) 1e-4 dB.
"""
import numpy as np
import pytest
import torch

from helpers import golden_setup, load_golden, mlp_params, scene_for
from oracle import pnr_oracle as O
from testdata import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from pixelnerf_amd import ops as _ops
    return _ops


def test_gen_rays_matches_reference(ops, dev):
    g = load_golden("neighbours")
    rays = ops.gen_rays(torch.from_numpy(g["rays_poses"]).to(dev), 20, 15, g["rays_focal"], 0.8, 1.8, c=g["rays_c"])
    np.testing.assert_allclose(rays.cpu().numpy(), g["rays_out"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", ["pool", "nopool"])
def test_pyramid_to_latent_matches_reference(ops, dev, name):
    g = load_golden("neighbours")
    stages = [t.to(dev) for t in synthetic.pyramid_stages(name)]
    nhwc, nchw = ops.pyramid_to_latent(stages)
    ref = g[f"pyr_{name}_latent"]
    assert nchw.shape == ref.shape and nhwc.shape == (ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1])
    np.testing.assert_allclose(nchw.cpu().numpy(), ref, rtol=0, atol=2e-6)
    assert torch.equal(nhwc.permute(0, 3, 1, 2), nchw)                       # both layouts carry the same bits
    np.testing.assert_array_equal(nchw[:, :64].cpu().numpy(), ref[:, :64])   # stage 0: scale 1 -> exact copy
    only, none = ops.pyramid_to_latent(stages, want_nchw=False)
    assert none is None and torch.equal(only, nhwc)


@pytest.mark.parametrize("shapes", [
    [(64, 9, 11), (64, 5, 6), (64, 3, 3)],                                      # 192 channels: groups do not tile 256 threads (generic sweep)
    [(64, 12, 40), (64, 6, 20), (128, 3, 10), (256, 2, 5), (512, 1, 3)],        # five stages, 1024 channels: shorter pixel runs (LDS)
    [(64, 1, 5), (128, 1, 1)],                                                  # a single row, a 1x1 stage
    [(64, 7, 1), (64, 4, 3)],                                                   # a single column
    [(64, 6, 8), (64, 13, 29), (128, 6, 8)],                                    # a stage LARGER than stage 0 (downsampled), and an equal-size one
    [(128, 33, 70)],                                                            # one stage: pure NCHW -> NHWC
])
def test_pyramid_to_latent_general_shapes(ops, dev, shapes):
    """pnr_pyramid_to_latent beyond the ResNet-34 shapes, against torch's own upsample + cat on the same device"""
    gen = torch.Generator().manual_seed(17)
    NV = 2
    stages = [torch.randn(NV, c, h, w, generator=gen).to(dev) for c, h, w in shapes]
    H0, W0 = shapes[0][1], shapes[0][2]
    ref = torch.cat([t if t.shape[2:] == (H0, W0) else torch.nn.functional.interpolate(t, (H0, W0), mode="bilinear", align_corners=True)
                     for t in stages], 1)
    nhwc, nchw = ops.pyramid_to_latent(stages)
    assert nchw.shape == ref.shape
    assert (nchw - ref).abs().max().item() <= 5e-6
    assert torch.equal(nhwc.permute(0, 3, 1, 2), nchw)
    only, none = ops.pyramid_to_latent(stages, want_nchw=False)
    assert none is None and torch.equal(only, nhwc)


def test_pyramid_to_latent_full_size_dtu(ops, dev):
    """BASELINE config (4) shapes: 3 views, 150x200 grid, 512 channels = 176 MiB per layout.  Properties:
    stage 0 copied exactly, align_corners => the four image corners of every stage are reproduced exactly,
    and the whole grid agrees with torch's own upsample on the same device."""
    stages = [t.to(dev) for t in synthetic.pyramid_stages("dtu")]
    nhwc, nchw = ops.pyramid_to_latent(stages)
    assert nhwc.shape == (3, 150, 200, 512)
    assert torch.equal(nchw[:, :64], stages[0])
    c0 = 64
    for t in stages[1:]:
        C = t.shape[1]
        for (yo, xo, ys, xs) in ((0, 0, 0, 0), (0, -1, 0, -1), (-1, 0, -1, 0), (-1, -1, -1, -1)):
            assert torch.equal(nchw[:, c0:c0 + C, yo, xo], t[:, :, ys, xs])
        c0 += C
    ref = torch.cat([torch.nn.functional.interpolate(t, (150, 200), mode="bilinear", align_corners=True) for t in stages], 1)
    assert (nchw - ref).abs().max().item() <= 5e-6
    assert torch.equal(nhwc.permute(0, 3, 1, 2), nchw)


def test_encoder_graph_replay_equals_eager_launches(dev):
    """eval-mode encodes under no_grad replay a HIP graph of the trunk + formatting pass (captured once per input shape): the
    eager launches' result, every call returns fresh tensors, a second shape gets its own graph, train mode stays eager"""
    from pixelnerf_amd.model.encoder import SpatialEncoder
    torch.manual_seed(5)
    enc = SpatialEncoder(pretrained=False, use_first_pool=False).to(dev).eval()
    imgs = [torch.rand(2, 3, 64, 64, device=dev) * 2 - 1 for _ in range(3)] + [torch.rand(1, 3, 48, 80, device=dev)]
    with torch.no_grad():
        enc.use_graph = False
        ref = [(enc(im).clone(), enc.latent_nhwc().clone(), enc.latent_scaling.clone()) for im in imgs]
        enc.use_graph = True
        got = []
        for im in imgs + imgs[:1]:
            lat = enc(im)
            got.append((lat, enc.latent_nhwc(), enc.latent_scaling.clone()))
    assert len(enc._graphs) == 2
    for (a, an, asc), (b, bn, bsc) in zip(got, ref + ref[:1]):
        # (the trunk's convolutions may run another MIOpen algorithm inside the capture: rounding-level differences)
        assert a.shape == b.shape and (a - b).abs().max().item() <= 1e-5 * max(1.0, b.abs().max().item())
        assert torch.equal(an.permute(0, 3, 1, 2), a) and torch.equal(asc, bsc)
    assert got[0][0].data_ptr() != got[1][0].data_ptr()  # fresh outputs: an earlier latent is not overwritten by the next encode
    enc.train()
    with torch.no_grad():
        before = len(enc._graphs)
        enc(imgs[0])
        assert len(enc._graphs) == before  # batch-norm statistics move in train mode: never captured
    type(enc).use_graph = True


def test_encoder_forward_uses_fused_formatting(dev):
    """SpatialEncoder.forward under no_grad (HIP formatting) == the torch formatting it replaces."""
    from pixelnerf_amd.model.encoder import SpatialEncoder
    torch.manual_seed(3)
    enc = SpatialEncoder(pretrained=False, use_first_pool=False).to(dev).eval()
    img = torch.rand(2, 3, 64, 64, device=dev) * 2 - 1
    with torch.no_grad():
        fused = enc(img).clone()
        nhwc = enc.latent_nhwc()
        scaling = enc.latent_scaling.clone()
    with torch.enable_grad():
        plain = enc(img).detach()
    assert fused.shape == (2, 512, 32, 32)
    assert (fused - plain).abs().max().item() <= 1e-5 * max(1.0, plain.abs().max().item())
    assert torch.equal(nhwc.permute(0, 3, 1, 2), fused)
    assert torch.equal(scaling, enc.latent_scaling)


@pytest.mark.parametrize("prec", ["f16", "f32"])
def test_render_views_equals_gen_rays_plus_render(ops, dev, prec):
    scene, meta = scene_for("mv_mini")
    sc = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev), scene["c"].to(dev),
                        scene["image_shape"], scene["NS"])
    pc = ops.pack_mlp({k: v.to(dev) for k, v in mlp_params(11).items()}, prec)
    pf = ops.pack_mlp({k: v.to(dev) for k, v in mlp_params(12).items()}, prec)
    W, H = 12, 10
    poses = torch.stack([torch.as_tensor(synthetic.pose_spherical(t, -20.0, 2.732)) for t in (20.0, 70.0, 150.0, 290.0)]).to(dev)
    R = 4 * W * H  # SB=2 objects x 2 target views each
    noise = {k: v.to(dev) for k, v in synthetic.make_noise(R, 16, 24, 8, seed=3).items()}
    kw = dict(depth_std=0.01, white_bkgd=True, lindisp=False, want_weights=True)
    a = ops.render_views(sc, pc, pf, poses, W, H, (59.7, 59.7), 1.2, 4.0, 16, 24, 8, noise, c=(6.0, 5.0), **kw)
    rays = ops.gen_rays(poses, W, H, (59.7, 59.7), 1.2, 4.0, c=(6.0, 5.0)).reshape(-1, 8)
    b = ops.render_forward(sc, pc, pf, rays, 16, 24, 8, noise, **kw)
    for p in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            assert torch.equal(a[p][k], b[p][k]), (p, k)
    with pytest.raises(ValueError):
        ops.render_views(sc, pc, pf, poses[:3], W, H, 59.7, 1.2, 4.0, 16, 0, 0, noise)


def test_eval_epilogue_matches_restatement(ops, dev):
    g = load_golden("neighbours")
    rs = np.random.RandomState(5)
    depth = rs.uniform(0.5, 2.5, (3, 300)).astype(np.float32)
    out = ops.eval_epilogue(torch.from_numpy(g["psnr_pred"]).to(dev), torch.from_numpy(depth).to(dev), 0.8, 1.8,
                            gt_rgb=torch.from_numpy(g["psnr_gt"]).to(dev))
    ref = O.eval_epilogue(g["psnr_pred"], depth, 0.8, 1.8, gt=g["psnr_gt"])
    np.testing.assert_array_equal(out["rgb"].cpu().numpy(), ref["rgb"])
    np.testing.assert_array_equal(out["rgb_u8"].cpu().numpy(), ref["rgb_u8"])
    np.testing.assert_array_equal(out["depth_norm"].cpu().numpy(), ref["depth_norm"])
    np.testing.assert_allclose(out["psnr"].cpu().numpy(), ref["psnr"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(out["psnr"].cpu().numpy(), g["psnr_out"], rtol=0, atol=1e-4)   # util.psnr of the reference
    again = ops.eval_epilogue(torch.from_numpy(g["psnr_pred"]).to(dev), gt_rgb=torch.from_numpy(g["psnr_gt"]).to(dev))
    assert torch.equal(again["sse"], out["sse"]) and "depth_norm" not in again   # deterministic reduction


def _training_batch(seed=9):
    rs = np.random.RandomState(seed)
    SB, NV, W, H, B = 3, 4, 20, 15, 64
    poses = torch.stack([torch.stack([torch.as_tensor(synthetic.pose_spherical(40.0 * o + 25.0 * v, -20.0 - 3 * v, 2.2))
                                      for v in range(NV)]) for o in range(SB)])
    images = torch.from_numpy(rs.uniform(-1, 1, (SB, NV, 3, H, W)).astype(np.float32))
    focal = torch.tensor([[41.5, 39.25], [38.0, 38.0], [45.0, 44.0]])
    c = torch.tensor([[10.75, 6.5], [10.0, 7.5], [9.0, 8.0]])
    return SB, NV, W, H, B, poses, images, focal, c, rs


def test_sample_training_rays_bbox(ops, dev):
    """bbox pixel sampling: the reference's util.bbox_sample draws (golden) through the HIP kernel give the
    reference's pixels; rays / colours equal the restated train.py glue (full ray map, then index)."""
    g = load_golden("neighbours")
    SB, NV, W, H, B, poses, images, focal, c, rs = _training_batch()
    # object 0 replays the golden draws (3 views -> ids < 3 are valid for NV = 4)
    bboxes = torch.from_numpy(np.concatenate([g["bbox_boxes"], [[1.0, 1.0, 5.0, 4.0]]]).astype(np.float32))[None].repeat(SB, 1, 1)
    ids = torch.from_numpy(g["bbox_ids"][:SB * B].reshape(SB, B))
    ux = torch.from_numpy(g["bbox_ux"][:SB * B].reshape(SB, B))
    uy = torch.from_numpy(g["bbox_uy"][:SB * B].reshape(SB, B))
    rays, gt = ops.sample_training_rays(poses.to(dev), images.to(dev), focal.to(dev), 0.8, 1.8, ids.to(dev), c=c.to(dev),
                                        bboxes=bboxes.to(dev), ux=ux.to(dev), uy=uy.to(dev))
    ref_rays, ref_gt = O.sample_training_rays(synthetic.gen_rays, poses, images, focal, 0.8, 1.8, ids, c=c, bboxes=bboxes, ux=ux, uy=uy)
    np.testing.assert_allclose(rays.cpu().numpy(), ref_rays.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_array_equal(gt.cpu().numpy(), ref_gt.numpy())
    # and the pixels implied by those rays are the reference's bbox_sample pixels
    pix = O.bbox_pixels(bboxes[0], ids.reshape(-1), ux.reshape(-1), uy.reshape(-1))
    np.testing.assert_array_equal(pix.numpy(), g["bbox_pix"][:SB * B])


def test_sample_training_rays_flat_indices(ops, dev):
    SB, NV, W, H, B, poses, images, focal, c, rs = _training_batch(11)
    ids = torch.from_numpy(rs.randint(0, NV * H * W, (SB, B)).astype(np.int64))
    ids[0, 0], ids[0, 1] = 0, NV * H * W - 1
    rays, gt = ops.sample_training_rays(poses.to(dev), images.to(dev), focal.to(dev), 0.8, 1.8, ids.to(dev))
    ref_rays, ref_gt = O.sample_training_rays(synthetic.gen_rays, poses, images, focal, 0.8, 1.8, ids)
    np.testing.assert_allclose(rays.cpu().numpy(), ref_rays.numpy(), rtol=0, atol=1e-6)
    np.testing.assert_array_equal(gt.cpu().numpy(), ref_gt.numpy())
    with pytest.raises(TypeError):
        ops.sample_training_rays(poses.to(dev), images.to(dev), focal.to(dev), 0.8, 1.8, ids.int().to(dev))
