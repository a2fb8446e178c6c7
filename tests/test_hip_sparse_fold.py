"""
GPU tests (-m gpu): the row-wise fold of fp32-class TRAINING on large grids (pnr_fold_latent_f32_rows, C ABI rev 8).

A training pass re-folds lin_z into the per-texel tables every step (the weights moved).  On a DTU-sized grid most texels are not
near any ray of the pass, so only the rows the pass reads are folded: every (view, point) is projected with the forward kernels'
own code, its four corner rows are marked, the marked rows are folded by the dense kernel's arithmetic.  What must hold:
  * the marked rows carry the bits of the dense fold (pnr_fold_latent_f32), everything else is left alone;
  * the training forward never reads an unmarked row -- checked with NaN in every unmarked row: one read of one of them, even with
    a zero weight (the padding points of the last tile), would surface as NaN in the output or in a kept operand image;
  * a training step through the row-wise fold ends in the bits of the step through the dense fold, eager and replayed from a HIP graph.
Reference lines: src/model/models.py:198-221 (projection + lookup), src/model/resnetfc.py:168-172 (lin_z).
"""
import pytest
import torch

from helpers import mlp_params
from testdata import synthetic

pytestmark = pytest.mark.gpu


def _scene(dev, name, hw=None):
    from pixelnerf_amd import ops
    s, meta = synthetic.make_scene(name)
    lat = s["latent"]
    if hw is not None:
        gen = torch.Generator().manual_seed(5)
        lat = torch.randn(lat.shape[0], 512, hw[0], hw[1], generator=gen) * 0.5
    sc = ops.make_scene(lat.to(dev), s["poses"].to(dev), s["focal"].to(dev), s["c"].to(dev), s["image_shape"], s["NS"])
    return sc, s, meta


# (scene, grid override, rays per object, samples): the full DTU grid (3 x 150 x 200); 2 objects x 2 views on 72 x 80 (ragged
# against the 256-row tiles and the 4096-texel compaction blocks); R K = 250 / 3 x 37 points: a last tile with padding points
@pytest.mark.parametrize("name,hw,n_rays,K", [("dtu", None, 128, 64), ("dtu", None, 25, 10), ("train_mv", (72, 80), 64, 96),
                                              ("train_mv", (72, 80), 3, 37)])
def test_rowwise_fold_is_the_dense_fold_on_the_rows_the_pass_reads(name, hw, n_rays, K):
    from pixelnerf_amd import ops
    dev = torch.device("cuda:0")
    sc, s, meta = _scene(dev, name, hw)
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    rays = synthetic.target_rays(meta, n_rays=n_rays).reshape(-1, 8).to(dev)
    gen = torch.Generator().manual_seed(9)
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], K, generator=gen).to(dev))
    dense = ops.fold_latent(sc, state, "f16x3")
    M = dense.shape[1] * dense.shape[2] * dense.shape[3]
    buf = torch.full_like(dense, float("nan"))
    ops.fold_latent_rows(sc, state, rays, z, buf)
    rows = buf.reshape(3, M, 512)
    marked = ~torch.isnan(rows[0, :, 0])
    n_marked = int(marked.sum())
    assert 0 < n_marked < M, (n_marked, M)  # the saving is real on these shapes
    pairs = rays.shape[0] * K * sc.NS
    assert n_marked <= 4 * pairs + sc.NS
    for b in range(3):  # a row is written in all three tables or in none, whole
        assert torch.equal(torch.isnan(rows[b]).any(dim=1), ~marked)
        assert torch.equal(rows[b][marked], dense.reshape(3, M, 512)[b][marked])
    # the training forward on NaN-padded tables: the bits of the dense-table forward, in the output and in every kept image
    pk = ops.pack_mlp(state, "f16x3", folded=True)
    out_d, saved_d = ops.eval_ray_samples_split_train(sc, pk, dense, rays, z)
    out_s, saved_s = ops.eval_ray_samples_split_train(sc, pk, buf, rays, z)
    assert torch.isfinite(out_s).all()
    assert torch.equal(out_s, out_d)
    kept = lambda sv: [sv.a[b] for b in range(5)] + [sv.n[b] for b in range(5)] + [sv.x5]  # noqa: E731
    for a, b in zip(kept(saved_s), kept(saved_d)):
        assert torch.equal(a, b)
    # a second pass with other samples into the SAME buffer: its rows are current, stale rows of the first pass stay untouched
    z2 = ops.sample_coarse(rays, torch.rand(rays.shape[0], K, generator=gen).to(dev))
    ops.fold_latent_rows(sc, state, rays, z2, buf)
    out_s2, _ = ops.eval_ray_samples_split_train(sc, pk, buf, rays, z2)
    out_d2, _ = ops.eval_ray_samples_split_train(sc, pk, dense, rays, z2)
    assert torch.equal(out_s2, out_d2)


def _training_setup(dev, name):
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util.conf import default_model_conf
    scene, meta = synthetic.make_scene(name)
    net = make_model(default_model_conf(), precision="f16x3").to(dev).train()
    net.mlp_coarse.load_state_dict(mlp_params(11))
    net.mlp_fine.load_state_dict(mlp_params(12))
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([float(lat.shape[-1]), float(lat.shape[-2])], device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=False).to(dev).train()
    rays = synthetic.target_rays(meta, n_rays=96).to(dev)
    noise = {k: v.to(dev) for k, v in synthetic.make_noise(rays.shape[0] * rays.shape[1], 64, 32, 16).items()}
    gt = torch.rand(rays.shape[0], rays.shape[1], 3, generator=torch.Generator().manual_seed(4)).to(dev)
    params = list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters())
    return net, rend, lat, rays, noise, gt, params


def test_training_step_through_the_rowwise_fold_ends_in_the_dense_bits(monkeypatch):
    """1 object x 3 views on the DTU grid (the rule picks the row-wise fold: 18-29 k (view, point) pairs against 90 k texels) against
    the same step with PIXELNERF_SPARSE_FOLD=0; then captured into a HIP graph and replayed."""
    dev = torch.device("cuda:0")
    net, rend, lat, rays, noise, gt, params = _training_setup(dev, "dtu")
    static_loss = torch.zeros((), device=dev)
    calls = {"rows": 0}
    from pixelnerf_amd import ops
    real = ops.fold_latent_rows

    def counting(*a, **k):
        calls["rows"] += 1
        return real(*a, **k)

    monkeypatch.setattr(ops, "fold_latent_rows", counting)

    def body():
        out = rend(net, rays, want_weights=True, _noise=noise)
        loss = ((out.coarse.rgb - gt) ** 2).mean() + ((out.fine.rgb - gt) ** 2).mean()
        for p in params:
            p.grad = None
        lat.grad = None
        loss.backward()
        static_loss.copy_(loss.detach())

    def snapshot():
        torch.cuda.synchronize()
        return [float(static_loss)] + [p.grad.clone() for p in params] + [lat.grad.clone()]

    monkeypatch.setenv("PIXELNERF_SPARSE_FOLD", "0")
    body()
    dense = snapshot()
    assert calls["rows"] == 0
    monkeypatch.delenv("PIXELNERF_SPARSE_FOLD")
    body()
    sparse = snapshot()
    assert calls["rows"] == 2  # coarse and fine network pass
    assert sparse[0] == dense[0]
    names = [n for n, _ in net.mlp_coarse.named_parameters()] + [n for n, _ in net.mlp_fine.named_parameters()] + ["latent"]
    bad = [(n, float((a - b).abs().max())) for n, a, b in zip(names, sparse[1:], dense[1:]) if not torch.equal(a, b)]
    assert not bad, bad
    assert float(lat.grad.abs().max()) > 0
    # the grid gradient of both passes accumulated in ONE buffer (large grids: the scatter's tiled form has one owner per element)
    # = the sum of per-pass buffers, bit for bit
    assert ops.latent_scatter_single_owner(net.scene(), rays.shape[0] * rays.shape[1], 64)
    monkeypatch.setattr(ops, "latent_scatter_single_owner", lambda *a, **k: False)
    body()
    two = snapshot()
    monkeypatch.undo()
    assert torch.equal(two[-1], sparse[-1]) and two[0] == sparse[0]
    # captured: memset + mark + compaction + fold are plain stream work on caller-owned memory
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        body()
    grads = [p.grad for p in params] + [lat.grad]
    for t in grads:
        t.zero_()
    static_loss.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert float(static_loss) == dense[0]
    bad = [(n, float((a - b).abs().max())) for n, a, b in zip(names, grads, dense[1:]) if not torch.equal(a, b)]
    assert not bad, bad


def test_rule_keeps_the_dense_fold_where_it_is_cheaper(monkeypatch):
    """config 5's grids (4 x 32 x 32 = 4096 texels) and passes with more (view, point) pairs than texels never take the row-wise fold"""
    dev = torch.device("cuda:0")
    from pixelnerf_amd import ops
    calls = {"rows": 0}
    real = ops.fold_latent_rows
    monkeypatch.setattr(ops, "fold_latent_rows", lambda *a, **k: (calls.__setitem__("rows", calls["rows"] + 1), real(*a, **k))[1])
    net, rend, lat, rays, noise, gt, params = _training_setup(dev, "train")
    out = rend(net, rays, want_weights=True, _noise=noise)
    assert calls["rows"] == 0 and torch.isfinite(out.fine.rgb).all()


def test_rowwise_fold_argument_validation():
    from pixelnerf_amd import ops, _lib
    dev = torch.device("cuda:0")
    sc, s, meta = _scene(dev, "train_mv", (72, 80))
    state = {k: v.to(dev) for k, v in mlp_params(11).items()}
    rays = synthetic.target_rays(meta, n_rays=8).reshape(-1, 8).to(dev)
    z = ops.sample_coarse(rays, torch.rand(rays.shape[0], 8, device=dev))
    with pytest.raises(_lib.PixelNerfHipError):
        ops.fold_latent_rows(sc, state, rays, z, torch.zeros(3, 4, 72, 80, 256, device=dev))
    with pytest.raises(_lib.PixelNerfHipError):
        ops.fold_latent_rows(sc, state, rays[:-1], z[:-1], torch.zeros(3, 4, 72, 80, 512, device=dev))  # R != SB * rays_per_obj
