"""
Counter-based random draws and in-kernel ray generation (SURVEY.md 7 "production mode", 8f rank 1).

The reference draws its sampling noise with torch.rand / randn launches (src/render/nerf.py:111,135,141,158) and builds
every ray on the host (src/util/util.py:238-276).  The seeded entries (pnr_render_forward_seeded, pnr_render_views) draw
inside the sampling kernels from Philox4x32-10 and regenerate rays from the camera where they are needed.

CONTRACT (what these tests pin):
  * the block function is Philox4x32-10: the three known-answer vectors of the Random123 distribution (CPU test, through
    pnr_philox_raw -- the same inline function the kernels call);
  * value i of draw d for global ray id g under seed s = word (i & 3) of Philox(counter = (g lo, g hi, i >> 2, d), key = s),
    top 24 bits / 2^24 (uniforms in [0,1)); normals: Box-Muller on word pairs of block i >> 1 of draw 3 -- checked
    against the numpy restatement oracle/philox.py;
  * seeded render == explicit-noise render fed ops.philox_noise(...) tensors, bit for bit;
  * the image does not depend on chunking / sharding when shards carry their ray-id placement;
  * in-kernel rays == pnr_gen_rays + explicit rays, bit for bit;
  * same seed -> same image, other seed -> other image; torch.manual_seed controls the renderer's seed.
"""
import ctypes

import numpy as np
import pytest
import torch

from helpers import golden_setup, mlp_params, scene_for
from oracle import philox as PH

KAT = [  # Random123 kat_vectors, philox4x32 with 10 rounds: counter, key, expected
    ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
    ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
    ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
]


def test_philox_known_answers():
    """CPU: the library's block function (host build of the inline device function) and the numpy checker."""
    from pixelnerf_amd import _lib
    lib = _lib.load()
    for ctr, key, want in KAT:
        c, k, o = (ctypes.c_uint32 * 4)(*ctr), (ctypes.c_uint32 * 2)(*key), (ctypes.c_uint32 * 4)()
        assert lib.pnr_philox_raw(c, k, o) == 0
        assert list(o) == want
        assert PH.philox4x32_10(np.array(ctr, np.uint32), np.array(key, np.uint32)).tolist() == want


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from pixelnerf_amd import ops as _ops
    return _ops


@pytest.mark.gpu
def test_draws_follow_the_counter_layout(ops, dev):
    seed = 0x0123456789ABCDEF
    R, per_obj, stride, off = 40, 10, 1000, 37  # 4 objects x 10 rays placed inside a 4 x 1000 ray set at offset 37
    nz = ops.philox_noise(R, 19, 13, 5, seed, dev, ray_id_offset=off, ray_id_stride=stride, rays_per_obj=per_obj)
    ids = np.array([(r // per_obj) * stride + r % per_obj + off for r in range(R)], np.uint64)
    assert np.array_equal(nz["u1"].cpu().numpy(), PH.uniforms(seed, ids, 0, 19))
    assert np.array_equal(nz["u2"].cpu().numpy(), PH.uniforms(seed, ids, 1, 8))
    assert np.array_equal(nz["u3"].cpu().numpy(), PH.uniforms(seed, ids, 2, 8))
    np.testing.assert_allclose(nz["n4"].cpu().numpy(), PH.normals(seed, ids, 5), rtol=0, atol=2e-5)  # fp32 log / cos / sqrt
    # 2^32 boundary of the ray id goes into the second counter word
    big = ops.philox_noise(2, 4, 0, 0, seed, dev, ray_id_offset=2 ** 32 - 1, ray_id_stride=5, rays_per_obj=5)
    assert np.array_equal(big["u1"].cpu().numpy(), PH.uniforms(seed, np.array([2 ** 32 - 1, 2 ** 32], np.uint64), 0, 4))


@pytest.mark.gpu
def test_draw_statistics(ops, dev):
    nz = ops.philox_noise(4096, 64, 128, 16, 7, dev)
    for k in ("u1", "u2", "u3"):
        u = nz[k].double()
        assert u.min().item() >= 0.0 and u.max().item() < 1.0
        assert abs(u.mean().item() - 0.5) < 2e-3 and abs(u.var().item() - 1.0 / 12.0) < 1e-3
    n = nz["n4"].double()
    assert torch.isfinite(n).all() and abs(n.mean().item()) < 1.5e-2 and abs(n.var().item() - 1.0) < 2e-2
    assert abs((n ** 4).mean().item() - 3.0) < 0.15  # kurtosis of a normal
    # rows (rays) and draws are decorrelated
    assert abs(torch.corrcoef(torch.stack([nz["u1"][:, 0], nz["u1"][:, 1]]))[0, 1].item()) < 0.05
    assert abs(torch.corrcoef(torch.stack([nz["u2"][:, 0], nz["u3"][:, 0]]))[0, 1].item()) < 0.05
    other = ops.philox_noise(4096, 64, 128, 16, 8, dev)
    assert not torch.equal(nz["u1"], other["u1"])


def _nets(ops, dev, name, fold=True, prec="f16"):
    s, _ = scene_for(name)
    sc = ops.make_scene(s["latent"].to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])
    st = [{k: v.to(dev) for k, v in mlp_params(seed).items()} for seed in (11, 12)]
    pk = [ops.pack_mlp(x, prec, folded=fold) for x in st]
    tabs = tuple(ops.fold_latent(sc, x, prec) for x in st) if (fold or prec == "f16x3") else None
    return sc, pk, tabs


@pytest.mark.gpu
@pytest.mark.parametrize("name,prec,fold", [("sn64_64_128", "f16", True), ("sn64_64_128", "f16", False), ("train_64_32", "bf16", True),
                                            ("mv_mini_lindisp", "f16", True), ("sn64_64_128", "f16x3", True)])
def test_seeded_render_equals_explicit_render_of_the_same_draws(ops, dev, name, prec, fold):
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    sc, pk, tabs = _nets(ops, dev, str(g["scene"]), fold, prec)
    r = rays.reshape(-1, 8).to(dev)
    R = r.shape[0]
    kw = dict(depth_std=float(g["depth_std"]), white_bkgd=bool(g["white_bkgd"]), lindisp=bool(g["lindisp"]), want_weights=True, tables=tabs)
    seed = 987654321987
    a = ops.render_forward(sc, pk[0], pk[1], r, Kc, Kf, Kfd, None, seed=seed, **kw)
    b = ops.render_forward(sc, pk[0], pk[1], r, Kc, Kf, Kfd, ops.philox_noise(R, Kc, Kf, Kfd, seed, dev, rays_per_obj=R // scene["SB"]), **kw)
    for p in a:
        for k in a[p]:
            assert torch.equal(a[p][k], b[p][k]), (p, k)
    assert torch.isfinite(a["coarse"]["rgb"]).all()
    c = ops.render_forward(sc, pk[0], pk[1], r, Kc, Kf, Kfd, None, seed=seed + 1, **kw)
    assert not torch.equal(a["coarse"]["rgb"], c["coarse"]["rgb"])


@pytest.mark.gpu
def test_seeded_image_is_independent_of_chunking_and_sharding(ops, dev):
    g, scene, meta, mc, mf, rays, noise = golden_setup("train_64_32")  # SB = 4 objects x 32 rays
    sc, pk, tabs = _nets(ops, dev, "train")
    B = rays.shape[1]
    r = rays.to(dev)
    kw = dict(white_bkgd=True, want_weights=True, tables=tabs, seed=42)
    whole = ops.render_forward(sc, pk[0], pk[1], r.reshape(-1, 8), 64, 32, 16, None, **kw)
    parts = []
    for lo, hi in ((0, 11), (11, 32)):  # dim-1 shards, as DataParallel(dim=1) / ShardedRenderWrapper cut them
        sh = r[:, lo:hi].reshape(-1, 8).contiguous()
        parts.append((lo, hi, ops.render_forward(sc, pk[0], pk[1], sh, 64, 32, 16, None, ray_id_offset=lo, ray_id_stride=B, **kw)))
    for p in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            full = whole[p][k].reshape((4, B) + tuple(whole[p][k].shape[1:]))
            cat = torch.cat([o[p][k].reshape((4, hi - lo) + tuple(o[p][k].shape[1:])) for lo, hi, o in parts], dim=1)
            assert torch.equal(full, cat), (p, k)


@pytest.mark.gpu
@pytest.mark.parametrize("fold", [True, False])
def test_render_views_generates_rays_and_draws_in_kernel(ops, dev, fold):
    """camera -> pixels in one call, nothing materialised == pnr_gen_rays + seeded render on explicit rays, bit for bit;
    a folded stream without its tables is refused."""
    from pixelnerf_amd import _lib
    from testdata import synthetic
    s, meta = scene_for("mv_mini")  # SB = 2 objects x NS = 2 views
    sc, pk, tabs = _nets(ops, dev, "mv_mini", fold)
    W, H = 12, 10
    poses = torch.stack([synthetic.pose_spherical(20.0 + 50.0 * i, -20.0, 2.7) for i in range(4)]).to(dev)  # 2 views per object
    kw = dict(c=(6.0, 5.0), white_bkgd=True, want_weights=True, tables=tabs)
    a = ops.render_views(sc, pk[0], pk[1], poses, W, H, (59.7, 58.1), 1.2, 4.0, 16, 24, 8, None, seed=5, **kw)
    rays = ops.gen_rays(poses, W, H, (59.7, 58.1), 1.2, 4.0, c=(6.0, 5.0)).reshape(-1, 8)
    b = ops.render_forward(sc, pk[0], pk[1], rays, 16, 24, 8, None, white_bkgd=True, want_weights=True, tables=tabs, seed=5)
    for p in a:
        for k in a[p]:
            assert torch.equal(a[p][k], b[p][k]), (p, k)
    # explicit noise through the camera entry
    nz = ops.philox_noise(rays.shape[0], 16, 24, 8, 5, dev, rays_per_obj=rays.shape[0] // 2)
    c = ops.render_views(sc, pk[0], pk[1], poses, W, H, (59.7, 58.1), 1.2, 4.0, 16, 24, 8, nz, **kw)
    assert torch.equal(a["fine"]["rgb"], c["fine"]["rgb"])
    if fold:
        with pytest.raises(_lib.PixelNerfHipError):
            ops.render_views(sc, pk[0], pk[1], poses, W, H, 59.7, 1.2, 4.0, 16, 24, 8, None, tables=None)


@pytest.mark.gpu
def test_renderer_seed_follows_torch_manual_seed(dev):
    from pixelnerf_amd.render import NeRFRenderer
    from test_api_gpu import build_net
    g, scene, meta, mc, mf, rays, noise = golden_setup("sn64_64_128")
    net = build_net(dev, scene)
    r = rays.to(dev)

    def run(seed, rng="philox"):
        torch.manual_seed(seed)
        rend = NeRFRenderer(n_coarse=64, n_fine=128, n_fine_depth=16, white_bkgd=True, rng=rng).to(dev).eval()
        with torch.no_grad():
            first = rend(net, r).fine.rgb.clone()
            second = rend(net, r).fine.rgb.clone()
        return first, second
    a1, a2 = run(3)
    b1, b2 = run(3)
    c1, _ = run(4)
    assert torch.equal(a1, b1) and torch.equal(a2, b2)      # reproducible under torch.manual_seed
    assert not torch.equal(a1, a2) and not torch.equal(a1, c1)  # fresh draws per call, other seed -> other draws
    t1, _ = run(3, rng="torch")
    t2, _ = run(3, rng="torch")
    assert torch.equal(t1, t2) and not torch.equal(t1, a1)  # the torch-generator mode is still there (reference draw order)
