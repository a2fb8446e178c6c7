"""
A procedural training scene for "trained-like weights" tests (no datasets or checkpoints exist offline): every object is an
analytically rendered, shaded sphere in front of a white background.  `sphere_targets` gives the ground-truth colour of any
ray; `fit` runs a short pixelNeRF training (both ResnetFCs and the feature grid, MSE coarse + MSE fine, Adam -- the
reference's train/train.py:199-215 objective) through whatever renderer it is handed.  The result is a network whose weights,
hidden-activation statistics and densities (a thin shell instead of fog) look like a trained checkpoint's rather than like
kaiming-init noise -- which is what the parity tests on trained-like weights need.  Test infrastructure, not product code.
"""
import numpy as np
import torch


def sphere_params(n_objs, seed=0):
    rs = np.random.RandomState(seed)
    centres = torch.from_numpy((rs.rand(n_objs, 3) * 0.4 - 0.2).astype(np.float32))
    radii = torch.from_numpy((0.45 + 0.25 * rs.rand(n_objs)).astype(np.float32))
    tints = torch.from_numpy((0.3 + 0.7 * rs.rand(n_objs, 3)).astype(np.float32))
    return centres, radii, tints


def sphere_targets(rays, centres, radii, tints):
    """rays (SB,B,8) -> rgb (SB,B,3): Lambert-ish shading 0.5 + 0.5 n, tinted per object; white where the ray misses"""
    o, d = rays[..., :3], rays[..., 3:6]
    oc = o - centres[:, None, :].to(rays.device)
    b = (oc * d).sum(-1)
    c = (oc * oc).sum(-1) - (radii[:, None].to(rays.device) ** 2)
    disc = b * b - c
    hit = disc > 0
    t = -b - torch.sqrt(disc.clamp_min(0))
    hit = hit & (t > rays[..., 6]) & (t < rays[..., 7])
    n = torch.nn.functional.normalize(oc + t.unsqueeze(-1) * d, dim=-1)
    shade = (0.5 + 0.5 * n) * tints[:, None, :].to(rays.device)
    return torch.where(hit.unsqueeze(-1), shade, torch.ones_like(shade))


def fit(net, renderer, latent, ray_pool, targets, steps=300, rays_per_obj=128, lr=5e-4, seed=0):
    """`steps` Adam steps on random `rays_per_obj`-ray batches of ray_pool (SB,N,8) / targets (SB,N,3); trains both MLPs and
    `latent` (a leaf tensor already installed as net.encoder.latent).  -> list of losses"""
    params = list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters()) + [latent]
    opt = torch.optim.Adam(params, lr=lr)
    gen = torch.Generator(device=ray_pool.device).manual_seed(seed)
    SB, N = ray_pool.shape[:2]
    losses = []
    for _ in range(steps):
        idx = torch.randint(0, N, (SB, rays_per_obj), device=ray_pool.device, generator=gen)
        rays = torch.gather(ray_pool, 1, idx.unsqueeze(-1).expand(-1, -1, 8))
        gt = torch.gather(targets, 1, idx.unsqueeze(-1).expand(-1, -1, 3))
        out = renderer(net, rays)
        loss = ((out.coarse.rgb - gt) ** 2).mean() + ((out.fine.rgb - gt) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return [float(v) for v in torch.stack(losses).cpu()]
