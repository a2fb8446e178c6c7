"""
GPU tests (-m gpu): NeRFRenderer around an ARBITRARY model callable is differentiable, like the reference's
(src/render/nerf.py:163-249 is plain autograd for any `model`): the HIP compositing / sampling kernels sit in the autograd graph
as pixelnerf_amd.autograd._CompositeFunction / _SampleFineFunction.  Checked against torch autograd on the CPU through the
oracle's restatement of the same lines (oracle.pnr_oracle.composite_from_rgbsigma, sample_fine, sample_fine_depth), including
the one position gradient of the reference: fine loss -> depth samples -> coarse depth (nerf.py:157-160,292).
Tolerance: everything is fp32 on both sides (the model is torch on both sides, the kernels restate fp32 arithmetic):
outputs 2e-6, gradients 1e-4 relative per tensor.
"""
import copy

import pytest
import torch

from oracle import pnr_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


class TinyField(torch.nn.Module):
    """An arbitrary radiance field with the reference's model call contract: (SB,B,3) points [+ viewdirs] -> (SB,B,4).
    Raw (sometimes negative) sigma on purpose: the relu is the renderer's (nerf.py:228)."""
    use_viewdirs = True

    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)

        def mlp():
            m = torch.nn.Sequential(torch.nn.Linear(6, 48), torch.nn.Softplus(), torch.nn.Linear(48, 48), torch.nn.Softplus(),
                                    torch.nn.Linear(48, 4))
            with torch.no_grad():
                for p in m.parameters():
                    p.copy_(torch.randn(p.shape, generator=g) * (0.6 if p.dim() > 1 else 0.1))
            return m
        self.coarse_net, self.fine_net = mlp(), mlp()

    def forward(self, xyz, coarse=True, viewdirs=None):
        h = (self.coarse_net if coarse else self.fine_net)(torch.cat([xyz, viewdirs], dim=-1))
        return torch.cat([torch.sigmoid(h[..., :3]), 2.0 * torch.sin(h[..., 3:4]) + 1.0], dim=-1)  # in [-1, 3]: mostly positive, some negative


def make_rays(SB, B, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.randn(SB, B, 3, generator=g) * 0.2 + torch.tensor([0.0, 0.0, -2.0])
    d = torch.nn.functional.normalize(torch.randn(SB, B, 3, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 1.0]), dim=-1)
    near = torch.full((SB, B, 1), 0.8)
    far = torch.full((SB, B, 1), 3.2) + torch.rand(SB, B, 1, generator=g) * 0.2
    return torch.cat([o, d, near, far], dim=-1)


def oracle_render(model, rays3, noise, Kc, Kf, Kfd, depth_std, white, lindisp, wc_for_sampling=None):
    """nerf.py:251-303 with torch ops on the CPU (the oracle's stage functions) around `model`."""
    SB = rays3.shape[0]
    rays = rays3.reshape(-1, 8)

    def composite(z, coarse):
        R, K = z.shape
        pts = (rays[:, None, :3] + z.unsqueeze(2) * rays[:, None, 3:6]).reshape(SB, -1, 3)
        vd = rays[:, None, 3:6].expand(-1, K, -1).reshape(SB, -1, 3)
        out = model(pts, coarse=coarse, viewdirs=vd).reshape(R, K, 4)
        return O.composite_from_rgbsigma(rays, z, out, white)

    z_c = O.sample_coarse(rays, noise["u1"], Kc, lindisp)
    wc, rgbc, depthc = composite(z_c, True)
    res = {"coarse": dict(weights=wc, rgb=rgbc, depth=depthc)}
    if Kf > 0:
        samps = [z_c]
        if Kf - Kfd > 0:
            w = wc.detach() if wc_for_sampling is None else wc_for_sampling
            samps.append(O.sample_fine(rays, w, noise["u2"], noise["u3"], Kc, lindisp))
        if Kfd > 0:
            samps.append(O.sample_fine_depth(rays, depthc, noise["n4"], depth_std))
        z_all, _ = torch.sort(torch.cat(samps, dim=-1), dim=-1)
        wf, rgbf, depthf = composite(z_all, False)
        res["fine"] = dict(weights=wf, rgb=rgbf, depth=depthf)
    return res


def loss_of(res, tgt):
    """a loss that touches every output: MSE on both rgb (train/train.py:199-215) + depth and weight terms"""
    loss = 0.0
    for i, p in enumerate(k for k in ("coarse", "fine") if k in res):
        r = res[p]
        rgb, depth, w = r["rgb"].reshape(-1, 3), r["depth"].reshape(-1), r["weights"].reshape(rgb_rows(r), -1)
        loss = loss + ((rgb - tgt["rgb"]) ** 2).mean() + 0.1 * ((depth - tgt["depth"]) ** 2).mean() + 0.05 * (w * tgt["w"][i][:, : w.shape[1]]).sum() / w.shape[0]
    return loss


def rgb_rows(r):
    return r["rgb"].reshape(-1, 3).shape[0]


@pytest.mark.parametrize("white,lindisp,Kfd", [(True, False, 8), (False, True, 4), (True, False, 0)])
def test_generic_model_training_matches_torch_autograd(dev, white, lindisp, Kfd):
    from pixelnerf_amd.render import NeRFRenderer
    SB, B, Kc, Kf, depth_std = 2, 48, 16, 16, 0.05
    R = SB * B
    rays = make_rays(SB, B, 5)
    g = torch.Generator().manual_seed(9)
    noise = {"u1": torch.rand(R, Kc, generator=g)}
    if Kf - Kfd > 0:
        noise["u2"], noise["u3"] = torch.rand(R, Kf - Kfd, generator=g), torch.rand(R, Kf - Kfd, generator=g)
    if Kfd > 0:
        noise["n4"] = torch.randn(R, Kfd, generator=g)
    tgt = {"rgb": torch.rand(R, 3, generator=g), "depth": torch.rand(R, generator=g) * 2 + 1,
           "w": [torch.randn(R, Kc + Kf, generator=g), torch.randn(R, Kc + Kf, generator=g)]}

    model_cpu = TinyField(3)
    model_gpu = copy.deepcopy(model_cpu).to(dev)
    renderer = NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=depth_std, white_bkgd=white, lindisp=lindisp,
                            eval_batch_size=700).to(dev).train()
    out = renderer(model_gpu, rays.to(dev), want_weights=True, _noise={k: v.to(dev) for k, v in noise.items()})
    got = {p: dict(rgb=out[p].rgb, depth=out[p].depth, weights=out[p].weights) for p in ("coarse", "fine")}
    assert got["fine"]["rgb"].requires_grad and got["coarse"]["depth"].requires_grad
    if Kfd > 0:  # the test is only meaningful for the position gradient when the depth samples are not clamped away
        d = got["coarse"]["depth"].detach().reshape(-1)
        assert float(((d > rays.reshape(-1, 8)[:, 6].to(dev) + 0.2) & (d < rays.reshape(-1, 8)[:, 7].to(dev) - 0.2)).float().mean()) > 0.5
    loss = loss_of(got, {"rgb": tgt["rgb"].to(dev), "depth": tgt["depth"].to(dev), "w": [t.to(dev) for t in tgt["w"]]})
    loss.backward()

    # the importance samples are a discontinuous function of the coarse weights (searchsorted bin): give the CPU pipeline the
    # HIP path's (detached) coarse weights for that one call so both sides sample the same bins
    ref = oracle_render(model_cpu, rays, noise, Kc, Kf, Kfd, depth_std, white, lindisp,
                        wc_for_sampling=got["coarse"]["weights"].detach().cpu().reshape(R, Kc))
    ref_loss = loss_of(ref, tgt)
    ref_loss.backward()

    for p in ("coarse", "fine"):
        for k in ("rgb", "depth", "weights"):
            a, b = got[p][k].detach().cpu().reshape(-1), ref[p][k].detach().reshape(-1)
            assert (a - b).abs().max() <= 2e-6 * max(1.0, float(b.abs().max())), (p, k, float((a - b).abs().max()))
    assert abs(float(loss) - float(ref_loss)) <= 1e-6 * max(1.0, abs(float(ref_loss)))
    worst = 0.0
    for (n, pg), (_, pc) in zip(model_gpu.named_parameters(), model_cpu.named_parameters()):
        assert pg.grad is not None and pc.grad is not None, n
        rel = float((pg.grad.cpu().double() - pc.grad.double()).norm() / (pc.grad.double().norm() + 1e-30))
        worst = max(worst, rel)
        assert rel <= 1e-4, f"{n}: relative gradient error {rel:.3e}"
    print(f"generic-model training (white={white}, lindisp={lindisp}, Kfd={Kfd}): worst relative gradient error {worst:.2e}")
    if Kfd > 0:
        # the depth-sample path is live: the fine loss alone reaches the COARSE network
        for p in model_gpu.parameters():
            p.grad = None
        out = renderer(model_gpu, rays.to(dev), _noise={k: v.to(dev) for k, v in noise.items()})
        ((out.fine.rgb.reshape(-1, 3) - tgt["rgb"].to(dev)) ** 2).mean().backward()
        gc = torch.cat([p.grad.reshape(-1) for p in model_gpu.coarse_net.parameters()])
        assert float(gc.abs().max()) > 0.0


def test_generic_model_under_no_grad_and_frozen_model_stay_plain(dev):
    """no_grad / a model without trainable parameters: the plain compositing kernel, no autograd node."""
    from pixelnerf_amd.render import NeRFRenderer
    rays = make_rays(1, 32, 1).to(dev)
    model = TinyField(4).to(dev)
    renderer = NeRFRenderer(n_coarse=8, n_fine=8, n_fine_depth=4).to(dev).eval()
    with torch.no_grad():
        a = renderer(model, rays)
    assert not a.fine.rgb.requires_grad
    for p in model.parameters():
        p.requires_grad_(False)
    torch.manual_seed(0)
    b = renderer(model, rays)
    assert not b.fine.rgb.requires_grad and b.fine.rgb.grad_fn is None
