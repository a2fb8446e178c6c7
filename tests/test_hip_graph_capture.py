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


@pytest.mark.parametrize("precision", ["f16x3", "f16"])
def test_training_step_is_capturable_and_replays_the_eager_bits(precision):
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util.conf import default_model_conf
    dev = torch.device("cuda:0")
    g, scene, meta, mc, mf, rays, noise = golden_setup("train_64_32")  # 4 objects x 32 rays, 64 + 32 (16 depth), 32x32 grids
    net = make_model(default_model_conf(), precision=precision).to(dev).train()
    net.mlp_coarse.load_state_dict(mlp_params(11))
    net.mlp_fine.load_state_dict(mlp_params(12))
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([32.0, 32.0], device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).to(dev).train()
    params = list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters())
    r = rays.to(dev)
    nz = {k: v.to(dev) for k, v in noise.items()}
    gt = torch.rand(4, 32, 3, device=dev)
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
        assert all(torch.equal(a, b) for a, b in zip(grads, eager[1:]))
    assert float(lat.grad.abs().max()) > 0
