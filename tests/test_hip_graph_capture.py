"""
GPU test (-m gpu): the training step of BASELINE config 5's shape inside a HIP graph (torch.cuda.CUDAGraph).

Every launch of the differentiable path -- the TRAIN forward, compositing, both backward chains, the weight-gradient launch, the
latent scatter with its segment pre-pass -- must be capturable: no allocation, no host synchronisation and no per-stream state
inside the library (the scatter's workspace comes from the caller since C ABI rev 7: a scratch keyed by stream was not found on
the capture stream and failed the capture).  The replayed graph must produce the bits of the eager step: parameter gradients are
fixed-order reductions, the grid gradient comes out of the LDS-slab scatter (pnr_bwd.hip).
"""
import pytest
import torch

from helpers import golden_setup, mlp_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("precision,name", [("f16x3", "train_64_32"), ("f16", "train_64_32"), ("f16x3", "srn_mini_64_128")])
def test_training_step_is_capturable_and_replays_the_eager_bits(precision, name):
    """train_64_32: 4 objects x 32 rays, one source view.  srn_mini_64_128: two source views -- the multi-view kernels' view-sum
    scratch is keyed by stream inside the library; the captured launches borrow the one the warm-up created (pnr_mlp.hip mv_scratch)."""
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util.conf import default_model_conf
    dev = torch.device("cuda:0")
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    Kc, Kf, Kfd = int(g["n_coarse"]), int(g["n_fine"]), int(g["n_fine_depth"])
    net = make_model(default_model_conf(), precision=precision).to(dev).train()
    net.mlp_coarse.load_state_dict(mlp_params(11))
    net.mlp_fine.load_state_dict(mlp_params(12))
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([float(lat.shape[-1]), float(lat.shape[-2])], device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, white_bkgd=bool(g["white_bkgd"]), lindisp=bool(g["lindisp"]),
                        depth_std=float(g["depth_std"])).to(dev).train()
    params = list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters())
    r = rays.to(dev)
    nz = {k: v.to(dev) for k, v in noise.items()}
    gt = torch.rand(r.shape[0], r.shape[1], 3, device=dev)
    static_loss = torch.zeros((), device=dev)

    def body():
        out = rend(net, r, want_weights=True, _noise=nz)
        loss = ((out.coarse.rgb - gt) ** 2).mean() + ((out.fine.rgb - gt) ** 2).mean()
        for p in params:
            p.grad = None
        lat.grad = None
        loss.backward()
        static_loss.copy_(loss.detach())

    body()
    torch.cuda.synchronize()
    eager = [float(static_loss)] + [p.grad.clone() for p in params] + [lat.grad.clone()]
    side = torch.cuda.Stream()  # warm-up on a side stream, as torch's capture recipe asks (allocator pools, lazy initialisations)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):  # captures on a stream of its own: nothing in the library may be keyed by the stream
        body()
    grads = [p.grad for p in params] + [lat.grad]  # the graph's static output tensors
    for _ in range(2):
        for t in grads:
            t.zero_()
        static_loss.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert float(static_loss) == eager[0]
        names = [n for n, _ in net.mlp_coarse.named_parameters()] + [n for n, _ in net.mlp_fine.named_parameters()] + ["latent"]
        bad = [(n, float((a - b).abs().max())) for n, a, b in zip(names, grads, eager[1:]) if not torch.equal(a, b)]
        assert not bad, bad
    assert float(lat.grad.abs().max()) > 0


@pytest.mark.parametrize("name", ["sn64_64_128", "dtu_mini_64_128"])
def test_render_call_is_capturable_and_replays_the_eager_bits(name):
    """inference: NeRFRenderer.forward (in-kernel draws off: explicit noise) captured once, replayed; single- and three-view scenes"""
    from test_api_gpu import build_net
    from pixelnerf_amd.render import NeRFRenderer
    dev = torch.device("cuda:0")
    g, scene, meta, mc, mf, rays, noise = golden_setup(name)
    net = build_net(dev, scene)
    rend = NeRFRenderer(n_coarse=int(g["n_coarse"]), n_fine=int(g["n_fine"]), n_fine_depth=int(g["n_fine_depth"]),
                        white_bkgd=bool(g["white_bkgd"]), lindisp=bool(g["lindisp"]), depth_std=float(g["depth_std"])).to(dev).eval()
    r = rays.to(dev)
    nz = {k: v.to(dev) for k, v in noise.items()}
    with torch.no_grad():
        eager = rend(net, r, want_weights=True, _noise=nz)
        e_rgb, e_w = eager.fine.rgb.clone(), eager.fine.weights.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            rend(net, r, want_weights=True, _noise=nz)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            out = rend(net, r, want_weights=True, _noise=nz)
        s_rgb, s_w = out.fine.rgb, out.fine.weights
        for _ in range(2):
            s_rgb.zero_(); s_w.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(s_rgb, e_rgb) and torch.equal(s_w, e_w)
