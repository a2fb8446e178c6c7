"""
CPU ORACLE for the pixelNeRF volume-rendering hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference algorithm.  It is
imported only by tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg; the
product path (pixel-nerf_amd/) never imports it and fails loudly when the HIP library is
missing.

Parity pin: the reference repository ships NO tests / golden vectors for this path
(SURVEY.md §4, §8c).  The oracle is therefore pinned against outputs of the reference
itself: `oracle/make_goldens.py` imports the unmodified reference code from
/root/reference (through the import stubs in oracle/ref_stubs/), runs it on the seeded
scenes of `testdata/synthetic.py` with pre-drawn noise, and freezes the outputs under
tests/golden/*.npz.  tests/test_oracle_vs_golden.py checks this restatement against those
fixtures to ~1e-6.

Every function cites the reference file:line it follows (paths relative to the reference
repository root).  All random draws are explicit inputs, in the reference's draw order
(src/render/nerf.py:111,135,141,158):
    u1 ~ U[0,1) (R, Kc)          coarse stratified jitter
    u2 ~ U[0,1) (R, Kf - Kfd)    inverse-CDF uniforms
    u3 ~ U[0,1) (R, Kf - Kfd)    in-bin jitter
    n4 ~ N(0,1) (R, Kfd)         depth-sample noise
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# model side
# --------------------------------------------------------------------------------------


def positional_encoding(x, num_freqs=6, freq_factor=1.5, include_input=True):
    """src/model/code.py:11-42.  x (N, d_in) -> (N, d_in*(2*num_freqs+1)).

    Layout: [x, sin(f0 x), sin(f0 x + pi/2), sin(f1 x), ...] with f_k = freq_factor * 2^k;
    the cosine is computed as a phase-shifted sine with fp32(pi/2) exactly as the reference
    does (code.py:24-26,38).
    """
    freqs = freq_factor * 2.0 ** torch.arange(0, num_freqs)  # code.py:15
    _freqs = torch.repeat_interleave(freqs, 2).view(1, -1, 1).to(x.device)  # code.py:21-23
    _phases = torch.zeros(2 * num_freqs)
    _phases[1::2] = math.pi * 0.5  # code.py:26-27
    _phases = _phases.view(1, -1, 1).to(x.device)
    embed = x.unsqueeze(1).repeat(1, num_freqs * 2, 1)  # code.py:37
    embed = torch.sin(torch.addcmul(_phases, embed, _freqs))  # code.py:38
    embed = embed.view(x.shape[0], -1)
    if include_input:
        embed = torch.cat((x, embed), dim=-1)  # code.py:40-41
    return embed


USE_GRID_SAMPLE = False  # baselines set this: call F.grid_sample exactly like the reference


def index_latent(latent, uv, image_shape, mode="bilinear", padding="border"):
    """src/model/encoder.py:80-109 + :161-163, with F.grid_sample(bilinear, border,
    align_corners=True) written out (or called directly when USE_GRID_SAMPLE, which is what
    the timed baselines use so that they run the reference's own ATen op).  mode / padding:
    encoder.py:27-28 `index_interp` / `index_padding`; anything but the shipped (bilinear, border)
    is the reference's own F.grid_sample call (encoder.py:100-108).

    latent (NV, C, Hl, Wl); uv (NV, N, 2) in source-image pixels; image_shape (W, H).
    Returns (NV, C, N).
    """
    NV, C, Hl, Wl = latent.shape
    # encoder.py:161-163  latent_scaling = [Wl, Hl] / ([Wl, Hl] - 1) * 2
    ls = torch.tensor([Wl, Hl], dtype=torch.float32)
    ls = ls / (ls - 1) * 2.0
    ls = ls.to(uv.device)
    scale = ls / image_shape.to(device=uv.device, dtype=torch.float32)  # encoder.py:98
    g = uv * scale - 1.0  # encoder.py:99
    if USE_GRID_SAMPLE or mode != "bilinear" or padding != "border":
        samples = torch.nn.functional.grid_sample(latent, g.unsqueeze(2), align_corners=True,
                                                  mode=mode, padding_mode=padding)
        return samples[:, :, :, 0]
    # grid_sample, align_corners=True: pix = (g + 1) / 2 * (size - 1)
    ix = ((g[..., 0] + 1) / 2) * (Wl - 1)
    iy = ((g[..., 1] + 1) / 2) * (Hl - 1)
    # padding_mode='border': clip coordinates into [0, size-1]
    ix = torch.clamp(ix, 0, Wl - 1)
    iy = torch.clamp(iy, 0, Hl - 1)
    # a NaN coordinate (0/0: a point at a source camera's centre line, models.py:206-209) comes out of ATen's
    # clip_coordinates as 0 -- pinned by tests/golden/adv_plane.npz (the reference run on such points): texel 0,
    # finite output, no NaN propagation.  +-inf (x/0) is ordinary border clamping.
    ix = torch.where(torch.isnan(ix), torch.zeros_like(ix), ix)
    iy = torch.where(torch.isnan(iy), torch.zeros_like(iy), iy)
    ix0 = torch.floor(ix)
    iy0 = torch.floor(iy)
    ix1 = ix0 + 1
    iy1 = iy0 + 1
    w_nw = (ix1 - ix) * (iy1 - iy)
    w_ne = (ix - ix0) * (iy1 - iy)
    w_sw = (ix1 - ix) * (iy - iy0)
    w_se = (ix - ix0) * (iy - iy0)

    def corner(iyc, ixc, w):
        inb = (ixc >= 0) & (ixc <= Wl - 1) & (iyc >= 0) & (iyc <= Hl - 1)
        ixl = ixc.clamp(0, Wl - 1).long()
        iyl = iyc.clamp(0, Hl - 1).long()
        flat = (iyl * Wl + ixl)  # (NV, N)
        vals = torch.gather(
            latent.reshape(NV, C, Hl * Wl), 2, flat.unsqueeze(1).expand(-1, C, -1)
        )  # (NV, C, N)
        return vals * (w * inb.to(w.dtype)).unsqueeze(1)

    out = corner(iy0, ix0, w_nw)
    out = out + corner(iy0, ix1, w_ne)
    out = out + corner(iy1, ix0, w_sw)
    out = out + corner(iy1, ix1, w_se)
    return out


def resnetfc_forward(p, zx, combine_inner_dims, d_latent=512, n_blocks=5, combine_layer=3,
                     return_hidden=False, combine_type="average"):
    """src/model/resnetfc.py:132-184 (ReLU activations, use_spade=False) with ResnetBlockFC.forward
    (resnetfc.py:53-62) inlined; combine_type = util.combine_interleaved's agg_type (util.py:461-471).

    p: dict of tensors with the reference state_dict names (lin_in.weight, ...).
    zx: (rows, d_latent + d_in).  combine_inner_dims = (NS, B).
    """
    z = zx[..., :d_latent]  # resnetfc.py:142
    x = zx[..., d_latent:]
    x = torch.nn.functional.linear(x, p["lin_in.weight"], p["lin_in.bias"])  # :147
    for b in range(n_blocks):
        if b == combine_layer:
            # util.combine_interleaved, src/util/util.py:461-471
            if not (len(combine_inner_dims) == 1 and combine_inner_dims[0] == 1):
                x = x.reshape(-1, *combine_inner_dims, *x.shape[1:])
                x = x.mean(dim=1) if combine_type == "average" else torch.max(x, dim=1)[0]  # util.py:465-468
        if d_latent > 0 and b < combine_layer:
            tz = torch.nn.functional.linear(z, p[f"lin_z.{b}.weight"], p[f"lin_z.{b}.bias"])
            x = x + tz  # resnetfc.py:175-180
        # ResnetBlockFC.forward resnetfc.py:55-62
        net = torch.nn.functional.linear(
            torch.relu(x), p[f"blocks.{b}.fc_0.weight"], p[f"blocks.{b}.fc_0.bias"]
        )
        dx = torch.nn.functional.linear(
            torch.relu(net), p[f"blocks.{b}.fc_1.weight"], p[f"blocks.{b}.fc_1.bias"]
        )
        x = x + dx
    out = torch.nn.functional.linear(torch.relu(x), p["lin_out.weight"], p["lin_out.bias"])
    if return_hidden:
        return out, x
    return out  # resnetfc.py:183


def repeat_interleave(t, repeats):
    """src/util/util.py:58-65."""
    out = t.unsqueeze(1).expand(-1, repeats, *t.shape[1:])
    return out.reshape(-1, *t.shape[1:])


def pixelnerf_forward(scene, mlp, xyz, viewdirs, return_hidden=False, combine_type="average"):
    """src/model/models.py:146-266 for the shipped configuration (use_encoder, use_xyz,
    normalize_z, use_code{6, 1.5, include_input}, use_viewdirs, not use_code_viewdirs,
    no global encoder).

    scene: dict(latent (SB*NS,C,Hl,Wl), poses (SB*NS,3,4) world->cam as stored by
    encode() models.py:112-114, focal (SB|1, 2) with fy already negated models.py:129-130,
    c (SB|1, 2), image_shape (2,)=(W,H), NS).
    xyz (SB, B, 3), viewdirs (SB, B, 3) -> (SB, B, 4) rgb sigma.
    """
    SB, B, _ = xyz.shape
    NS = scene["NS"]
    poses = scene["poses"]
    xyz = repeat_interleave(xyz, NS)  # models.py:161
    xyz_rot = torch.matmul(poses[:, None, :3, :3], xyz.unsqueeze(-1))[..., 0]  # :162-164
    xyz_cam = xyz_rot + poses[:, None, :3, 3]  # :165
    z_feature = xyz_rot.reshape(-1, 3)  # :169-171 (normalize_z)
    z_feature = positional_encoding(z_feature)  # :180-182
    vd = viewdirs.reshape(SB, B, 3, 1)  # :188
    vd = repeat_interleave(vd, NS)
    vd = torch.matmul(poses[:, None, :3, :3], vd).reshape(-1, 3)  # :190-193
    z_feature = torch.cat((z_feature, vd), dim=1)  # :194-196
    uv = -xyz_cam[:, :, :2] / xyz_cam[:, :, 2:]  # :206
    focal, c = scene["focal"], scene["c"]
    uv = uv * repeat_interleave(focal.unsqueeze(1), NS if focal.shape[0] > 1 else 1)  # :207-209
    uv = uv + repeat_interleave(c.unsqueeze(1), NS if c.shape[0] > 1 else 1)  # :210-212
    latent = index_latent(scene["latent"], uv, scene["image_shape"])  # :213-215
    latent = latent.transpose(1, 2).reshape(-1, latent.shape[1])  # :219-221
    mlp_input = torch.cat((latent, z_feature), dim=-1)  # :227
    out = resnetfc_forward(mlp, mlp_input, (NS, B), return_hidden=return_hidden, combine_type=combine_type)  # :242-255
    hidden = None
    if return_hidden:
        out, hidden = out
    out = out.reshape(-1, B, 4)
    rgb = torch.sigmoid(out[..., :3])  # :260-263
    sigma = torch.relu(out[..., 3:4])
    res = torch.cat([rgb, sigma], dim=-1).reshape(SB, B, -1)
    if return_hidden:
        return res, hidden.reshape(SB, B, -1)  # residual stream in front of lin_out
    return res


def resnetfc_forward_general(p, zx, combine_inner_dims, d_in, d_latent, d_hidden, n_blocks, combine_layer=1000,
                             combine_type="average", beta=0.0, use_spade=False, kink_margin=None):
    """src/model/resnetfc.py:132-184 for ANY constructor arguments (resnetfc.py:66-130): Softplus(beta) instead of ReLU when
    beta > 0 (:126-129, :43-46), SPADE modulation (:176-179), d_in == 0 (:149), d_latent == 0 (:144-145), a combine layer that is
    never reached.  ResnetBlockFC.forward (:53-62) inlined (ResnetFC only builds equal-width blocks: no shortcut)."""
    F = torch.nn.functional
    if beta > 0:
        act = lambda t: F.softplus(t, beta=beta)
    else:
        def act(t):
            # test aid (not part of the reference): kink_margin (a list) collects min |t| / rms(t) over every ReLU input -- how
            # close this input comes to a ReLU kink, where two arithmetic classes may legitimately pick different branches
            if kink_margin is not None:
                kink_margin.append(float(t.detach().abs().min() / (t.detach().pow(2).mean().sqrt() + 1e-30)))
            return torch.relu(t)
    if d_latent > 0:
        z, x = zx[..., :d_latent], zx[..., d_latent:]  # :142-143
    else:
        x = zx
    if d_in > 0:
        x = F.linear(x, p["lin_in.weight"], p["lin_in.bias"])  # :147
    else:
        x = torch.zeros(d_hidden, device=zx.device)  # :149
    for b in range(n_blocks):
        if b == combine_layer:  # :153-172 -> util.combine_interleaved, util.py:461-471
            if not (len(combine_inner_dims) == 1 and combine_inner_dims[0] == 1):
                x = x.reshape(-1, *combine_inner_dims, *x.shape[1:])
                x = x.mean(dim=1) if combine_type == "average" else torch.max(x, dim=1)[0]
        if d_latent > 0 and b < combine_layer:
            tz = F.linear(z, p[f"lin_z.{b}.weight"], p[f"lin_z.{b}.bias"])  # :175
            if use_spade:
                sz = F.linear(z, p[f"scale_z.{b}.weight"], p[f"scale_z.{b}.bias"])  # :177
                x = sz * x + tz  # :178
            else:
                x = x + tz  # :180
        net = F.linear(act(x), p[f"blocks.{b}.fc_0.weight"], p[f"blocks.{b}.fc_0.bias"])  # :55
        x = x + F.linear(act(net), p[f"blocks.{b}.fc_1.weight"], p[f"blocks.{b}.fc_1.bias"])  # :56-62
    return F.linear(act(x), p["lin_out.weight"], p["lin_out.bias"])  # :183


def pixelnerf_forward_general(scene, mlp, xyz, viewdirs, conf, global_latent=None):
    """src/model/models.py:146-266 for ANY model conf (models.py:22-65): `conf` is the reference's model conf as a nested dict
    (testdata.synthetic.variant_model_conf), `mlp` the state dict of the network to run (mlp_coarse / mlp_fine) whose shape is
    conf["mlp_coarse"].  global_latent (SB*NS, G): what ImageEncoder.forward leaves in `global_encoder.latent` (encoder.py:220)."""
    g = lambda k, d: conf.get(k, d)
    use_encoder, use_xyz, normalize_z = g("use_encoder", True), g("use_xyz", False), g("normalize_z", True)
    use_code, use_code_viewdirs, use_viewdirs = g("use_code", False), g("use_code_viewdirs", True), g("use_viewdirs", False)
    code = conf.get("code", {})
    enc = lambda t: positional_encoding(t, code.get("num_freqs", 6), code.get("freq_factor", math.pi), code.get("include_input", True))
    SB, B, _ = xyz.shape
    NS = scene["NS"]
    poses = scene["poses"]
    xyz = repeat_interleave(xyz, NS)  # :161
    xyz_rot = torch.matmul(poses[:, None, :3, :3], xyz.unsqueeze(-1))[..., 0]  # :162-164
    xyz_cam = xyz_rot + poses[:, None, :3, 3]  # :165
    src = xyz_rot if normalize_z else xyz_cam  # :169-179
    z_feature = src.reshape(-1, 3) if use_xyz else -src[..., 2].reshape(-1, 1)
    if use_code and not use_code_viewdirs:
        z_feature = enc(z_feature)  # :180-182
    if use_viewdirs:
        vd = repeat_interleave(viewdirs.reshape(SB, B, 3, 1), NS)  # :188-189
        vd = torch.matmul(poses[:, None, :3, :3], vd).reshape(-1, 3)  # :190-193
        z_feature = torch.cat((z_feature, vd), dim=1)  # :194-196
    if use_code and use_code_viewdirs:
        z_feature = enc(z_feature)  # :198-200
    mlp_input = z_feature  # :202
    d_latent = 0
    if use_encoder:
        uv = -xyz_cam[:, :, :2] / xyz_cam[:, :, 2:]  # :206
        focal, c = scene["focal"], scene["c"]
        uv = uv * repeat_interleave(focal.unsqueeze(1), NS if focal.shape[0] > 1 else 1)  # :207-209
        uv = uv + repeat_interleave(c.unsqueeze(1), NS if c.shape[0] > 1 else 1)  # :210-212
        enc_conf = conf.get("encoder", {})
        latent = index_latent(scene["latent"], uv, scene["image_shape"], enc_conf.get("index_interp", "bilinear"),
                              enc_conf.get("index_padding", "border"))  # :213-215
        d_latent = latent.shape[1]
        latent = latent.transpose(1, 2).reshape(-1, d_latent)  # :219-221
        mlp_input = torch.cat((latent, z_feature), dim=-1)  # :227
    if g("use_global_encoder", False):
        reps = mlp_input.shape[0] // global_latent.shape[0]  # :231-233
        mlp_input = torch.cat((repeat_interleave(global_latent, reps), mlp_input), dim=-1)  # :234-235
        d_latent += global_latent.shape[1]
    m = conf["mlp_coarse"]
    out = resnetfc_forward_general(mlp, mlp_input, (NS, B), mlp_input.shape[1] - d_latent, d_latent, m.get("d_hidden", 128),
                                   m.get("n_blocks", 5), m.get("combine_layer", 1000), m.get("combine_type", "average"),
                                   m.get("beta", 0.0), m.get("use_spade", False))  # :242-255
    out = out.reshape(-1, B, 4)  # :258
    return torch.cat([torch.sigmoid(out[..., :3]), torch.relu(out[..., 3:4])], dim=-1).reshape(SB, B, -1)  # :260-265


# --------------------------------------------------------------------------------------
# renderer side
# --------------------------------------------------------------------------------------


def _z_from_steps(rays, z_steps, lindisp):
    near, far = rays[:, -2:-1], rays[:, -1:]
    if not lindisp:
        return near * (1 - z_steps) + far * z_steps  # nerf.py:113
    return 1 / (1 / near * (1 - z_steps) + 1 / far * z_steps)  # nerf.py:115


def sample_coarse(rays, u1, n_coarse, lindisp=False):
    """src/render/nerf.py:98-118.  rays (R,8), u1 (R,Kc) -> (R,Kc)."""
    step = 1.0 / n_coarse
    R = rays.shape[0]
    z_steps = torch.linspace(0, 1 - step, n_coarse, device=rays.device)  # :109
    z_steps = z_steps.unsqueeze(0).repeat(R, 1)
    z_steps = z_steps + u1 * step  # :111
    return _z_from_steps(rays, z_steps, lindisp)


def sample_fine(rays, weights, u2, u3, n_coarse, lindisp=False):
    """src/render/nerf.py:120-148.  weights (R,Kc) detached; u2,u3 (R,Kf-Kfd)."""
    weights = weights.detach() + 1e-5  # :130
    pdf = weights / torch.sum(weights, -1, keepdim=True)  # :131
    cdf = torch.cumsum(pdf, -1)  # :132
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)  # :133
    inds = torch.searchsorted(cdf, u2.contiguous(), right=True).float() - 1.0  # :138
    inds = torch.clamp_min(inds, 0.0)  # :139
    z_steps = (inds + u3) / n_coarse  # :141
    return _z_from_steps(rays, z_steps, lindisp)


def sample_fine_depth(rays, depth, n4, depth_std):
    """src/render/nerf.py:150-161.  depth (R,), n4 (R,Kfd)."""
    z = depth.unsqueeze(1).repeat((1, n4.shape[1]))
    z = z + n4 * depth_std  # :158
    z = torch.max(torch.min(z, rays[:, -1:]), rays[:, -2:-1])  # :160
    return z


def composite_from_rgbsigma(rays, z_samp, out, white_bkgd):
    """src/render/nerf.py:178-182 (deltas) and :223-249 (alpha compositing).
    z_samp (R,K), out (R,K,4) -> weights (R,K), rgb (R,3), depth (R)."""
    deltas = z_samp[:, 1:] - z_samp[:, :-1]
    delta_inf = rays[:, -1:] - z_samp[:, -1:]  # :181 (far - z_last, not 1e10)
    deltas = torch.cat([deltas, delta_inf], -1)
    rgbs = out[..., :3]
    sigmas = out[..., 3]
    alphas = 1 - torch.exp(-deltas * torch.relu(sigmas))  # :228
    alphas_shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    T = torch.cumprod(alphas_shifted, -1)  # :234
    weights = alphas * T[:, :-1]  # :235
    rgb_final = torch.sum(weights.unsqueeze(-1) * rgbs, -2)  # :239
    depth_final = torch.sum(weights * z_samp, -1)  # :240
    if white_bkgd:
        pix_alpha = weights.sum(dim=1)
        rgb_final = rgb_final + 1 - pix_alpha.unsqueeze(-1)  # :241-244
    return weights, rgb_final, depth_final


def composite(scene, mlp, rays, z_samp, sb, white_bkgd, sigma_noise=None, eval_batch_size=None):
    """src/render/nerf.py:163-249.  eval_batch_size=None: one model call (chunking does not change results: the model is
    pointwise); an integer: the reference's chunk loop (nerf.py:190-216: points split along the per-object axis into
    (eval_batch_size - 1) // sb + 1 points per model call) -- the execution shape bench.py times as the reference's.
    sigma_noise (R,K), already scaled by noise_std: the training-time `sigmas + randn_like(sigmas) * noise_std` of
    nerf.py:225-226 with the draw made explicit."""
    R, K = z_samp.shape
    points = rays[:, None, :3] + z_samp.unsqueeze(2) * rays[:, None, 3:6]  # :185
    points = points.reshape(sb, -1, 3)  # :193-195
    viewdirs = rays[:, None, 3:6].expand(-1, K, -1).reshape(sb, -1, 3)  # :204-206
    if eval_batch_size is None:
        out = pixelnerf_forward(scene, mlp, points, viewdirs)
    else:
        chunk = (int(eval_batch_size) - 1) // sb + 1  # :195
        out = torch.cat([pixelnerf_forward(scene, mlp, p, d)  # :207-213
                         for p, d in zip(torch.split(points, chunk, dim=1), torch.split(viewdirs, chunk, dim=1))], dim=1)  # :218
    out = out.reshape(R, K, -1)  # :219
    if sigma_noise is not None:  # :225-226
        out = torch.cat([out[..., :3], out[..., 3:4] + sigma_noise.unsqueeze(-1)], dim=-1)
    return composite_from_rgbsigma(rays, z_samp, out, white_bkgd) + (out,)


def render(scene, mlp_coarse, mlp_fine, rays, noise, n_coarse, n_fine, n_fine_depth,
           depth_std=0.01, white_bkgd=False, lindisp=False, detach_depth=False, sigma_noise=None, eval_batch_size=None,
           sampling_weights=None):
    """src/render/nerf.py:251-303.  rays (SB, B, 8); noise = dict(u1,u2,u3,n4) (missing keys
    allowed when the corresponding stage is skipped).  Returns a nested dict
    {coarse:{rgb,depth,weights,z,rgbsigma}, fine:{...}}; `fine` absent when n_fine == 0.
    mlp_fine=None falls back to mlp_coarse (models.py:242).
    sampling_weights (R,Kc): coarse weights to draw the importance samples from instead of this run's own (which are detached
    there anyway, nerf.py:288) -- a comparison with another implementation hands over ITS coarse weights so that both sides land
    in the same cdf bins (the bin index is a discontinuous function of the weights: a 1-ulp difference can move a sample)."""
    assert rays.dim() == 3  # :269
    SB = rays.shape[0]
    rays = rays.reshape(-1, 8)
    z_coarse = sample_coarse(rays, noise["u1"], n_coarse, lindisp)  # :273
    # sigma_noise: (noise of the coarse pass (R,Kc), of the fine pass (R,Kc+Kf)), nerf.py:225-226 (training, noise_std > 0)
    wc, rgbc, depthc, outc = composite(scene, mlp_coarse, rays, z_coarse, SB, white_bkgd,
                                       None if sigma_noise is None else sigma_noise[0], eval_batch_size)

    def fmt(w, rgb, depth, z, out):
        return dict(
            rgb=rgb.reshape(SB, -1, 3), depth=depth.reshape(SB, -1),
            weights=w.reshape(SB, -1, w.shape[-1]), z=z.reshape(SB, -1, z.shape[-1]),
            rgbsigma=out.reshape(SB, -1, out.shape[-2], 4),
        )

    ret = dict(coarse=fmt(wc, rgbc, depthc, z_coarse, outc))
    if n_fine > 0:  # using_fine, :87,:284
        all_samps = [z_coarse]
        if n_fine - n_fine_depth > 0:
            all_samps.append(
                sample_fine(rays, wc.detach() if sampling_weights is None else sampling_weights, noise["u2"], noise["u3"], n_coarse, lindisp)
            )  # :286-289
        if n_fine_depth > 0:
            # the reference passes the NON-detached coarse depth here (:292); detach_depth=True
            # removes that one position-gradient path (what the HIP backward implements so far)
            all_samps.append(sample_fine_depth(rays, depthc.detach() if detach_depth else depthc,
                                               noise["n4"], depth_std))  # :290-293
        z_combine = torch.cat(all_samps, dim=-1)
        z_sorted, _ = torch.sort(z_combine, dim=-1)  # :294-295
        mf = mlp_fine if mlp_fine is not None else mlp_coarse
        wf, rgbf, depthf, outf = composite(scene, mf, rays, z_sorted, SB, white_bkgd,
                                           None if sigma_noise is None else sigma_noise[1], eval_batch_size)  # :296
        ret["fine"] = fmt(wf, rgbf, depthf, z_sorted, outf)
    return ret


def psnr(pred, target):
    """src/util/util.py:474-481."""
    mse = ((pred - target) ** 2).mean().item()
    if mse == 0:
        return float("inf")
    return -10 * math.log10(mse)


# ---------------------------------------------------------------- neighbours of the path


def encoder_format(stages, upsample_interp="bilinear"):
    """src/model/encoder.py:150-163: every stage upsampled to stage 0's size (align_corners=True) and
    concatenated on channels -> (latent NCHW, latent_scaling)."""
    sz = stages[0].shape[-2:]
    lat = torch.cat([F.interpolate(t, sz, mode=upsample_interp, align_corners=True) for t in stages], dim=1)
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32)
    return lat, ls / (ls - 1) * 2.0


def eval_epilogue(rgb, depth, z_near, z_far, gt=None):
    """eval/eval.py:283-290 (depth normalisation, clamp), :302 (uint8 = trunc(x*255)), :327-329
    (skimage compare_psnr, data_range=1: fp64 mean squared error) -- numpy restatement."""
    rgb = np.clip(np.asarray(rgb, np.float32), 0.0, 1.0)
    out = {"rgb": rgb, "rgb_u8": (rgb * 255).astype(np.uint8),
           "depth_norm": (np.asarray(depth, np.float32) - np.float32(z_near)) / (np.float32(z_far) - np.float32(z_near))}
    if gt is not None:
        d = rgb.astype(np.float64) - np.asarray(gt, np.float32).astype(np.float64)
        mse = (d * d).reshape(rgb.shape[0], -1).mean(axis=1)
        out["psnr"] = 10.0 * np.log10(1.0 / mse)
    return out


def bbox_pixels(bboxes, image_ids, ux, uy):
    """src/util/util.py:220-235 with the draws (randint, rand, rand) made explicit -> (n,3) [image, y, x]."""
    pb = bboxes[image_ids]
    x = (ux * (pb[:, 2] + 1 - pb[:, 0]) + pb[:, 0]).long()
    y = (uy * (pb[:, 3] + 1 - pb[:, 1]) + pb[:, 1]).long()
    return torch.stack((image_ids, y, x), dim=-1)


def sample_training_rays(gen_rays, poses, images, focal, z_near, z_far, ids, c=None, bboxes=None, ux=None, uy=None):
    """train/train.py:143-182: per object, the full (NV,H,W,8) ray map (`gen_rays` = a restatement of
    util.gen_rays) and the (NV,H,W,3) colour map indexed at the drawn pixels."""
    SB, NV = poses.shape[:2]
    H, W = images.shape[-2:]
    all_rays, all_gt = [], []
    for o in range(SB):
        cam_rays = gen_rays(poses[o], W, H, focal[o], z_near, z_far, c=None if c is None else c[o])
        rgb_all = (images[o] * 0.5 + 0.5).permute(0, 2, 3, 1).contiguous().reshape(-1, 3)
        if bboxes is not None:
            pix = bbox_pixels(bboxes[o], ids[o], ux[o], uy[o])
            pix_inds = pix[..., 0] * H * W + pix[..., 1] * W + pix[..., 2]
        else:
            pix_inds = ids[o]
        all_gt.append(rgb_all[pix_inds])
        all_rays.append(cam_rays.reshape(-1, 8)[pix_inds])
    return torch.stack(all_rays), torch.stack(all_gt)
