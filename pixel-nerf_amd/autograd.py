"""
Differentiable NeRFRenderer.forward (training; BASELINE config 5, train/train.py:199-215).

forward  : the inference kernels, with the network launches replaced by their training form
           (`pnr_eval_ray_samples_train`: same kernel + 16-bit dumps of every linear's input).
backward : HIP kernels only (no library GEMM, no torch matmul) --
             pnr_composite_backward   d(rgb, depth, weights)      -> d(per-point rgb sigma)
             pnr_mlp_backward         fused data-gradient chain   -> per-layer output gradients dY,
                                      d z_lat = sum_b dY_b W_z[b] and d(code) = dY W_in (transposed weight streams)
             pnr_latent_scatter       d(interpolated latent)      -> d(feature grid)
             pnr_weight_grad_batched  dW = dY^T X, db = sum dY    from the 16-bit dumps (MFMA, fp32 acc), one launch
             pnr_lin_out_grad         lin_out's 4 x 512 weight gradient
             pnr_position_backward    d(network inputs)           -> d(sample positions z)
At the default precision "f16x3" (fp32-class) the network calls are replaced by their split-operand forms, fused the same way:
             pnr_eval_ray_samples_split_train   the fp32-class inference kernel in its training instantiation (operand images
                                                + relu masks kept, lin_z through the folded fp32 tables)
             pnr_mlp_backward_split             one launch for all transposed products, one batched split-operand
                                                weight-gradient launch -> all 30 gradients, d z_lat, d(code)
(FUSED_SPLIT_TRAINING = False: the same arithmetic as one split-operand GEMM per layer; precision "f32": exact fp32 MFMA.)
Gradients flow to every ResnetFC parameter of both networks and to `encoder.latent` (hence into
the ResNet-34 through PyTorch autograd), including the reference's one position-gradient path:
fine loss -> positions of the n_fine_depth samples (compositing deltas/depth, positional code,
projection + bilinear lookup) -> sort permutation -> clamp -> coarse depth (nerf.py:157-160,292)
-> coarse network.  Coarse and importance samples carry no gradient in the reference either
(rays are inputs, importance weights are detached, nerf.py:288).
"""

import torch

from . import ops

def _param_names():
    names = ["lin_in", "lin_out"] + [f"blocks.{b}.fc_{j}" for b in range(5) for j in (0, 1)] + [f"lin_z.{b}" for b in range(3)]
    return [n + s for n in names for s in (".weight", ".bias")]


PARAM_NAMES = _param_names()

# precision "f16x3": training runs FUSED -- the split-operand inference kernel in its training instantiation (one launch per
# network pass, operand images + relu masks kept), one launch for all transposed products of the backward, one batched
# split-operand weight-gradient launch.  False = the GEMM-per-layer form (~120 split-operand GEMM launches per step: same
# arithmetic class, the A/B twin and the yardstick of the gradient tests)
FUSED_SPLIT_TRAINING = True


def _sigma_noise(rgbs, cfg):
    """nerf.py:225-226 (training only, noise_std > 0): sigmas = sigmas + randn_like(sigmas) * noise_std, drawn from torch's
    generator after each network pass like the reference.  Under a sharding wrapper (cfg["ray_id_stride"] = rays per object of
    the WHOLE batch, cfg["ray_id_offset"] = this shard's first ray) the draw is made for the whole batch and this shard's rows
    are cut out of it: ranks / devices that share a seed then add to every ray exactly the noise a single process would have
    added (identical draws for different rays on different ranks would correlate the shards).
    -> (rgb | noisy sigma, mask of points whose own sigma was positive -- the relu' the compositing backward can no longer
    read off the noisy value) or (rgbs, None)"""
    std = cfg.get("noise_std", 0.0)
    if not std > 0.0:
        return rgbs, None
    live = (rgbs[..., 3] > 0).float()
    noisy = rgbs.clone()
    R, K = rgbs.shape[:2]
    stride, lo = int(cfg.get("ray_id_stride", 0) or 0), int(cfg.get("ray_id_offset", 0) or 0)
    SB = max(int(cfg.get("num_objs", 1) or 1), 1)
    if stride > 0 and R % SB == 0 and lo + R // SB <= stride and R // SB != stride:
        b = R // SB
        n = torch.randn((SB, stride, K), dtype=rgbs.dtype, device=rgbs.device)[:, lo:lo + b].reshape(R, K)
    else:
        n = torch.randn((R, K), dtype=rgbs.dtype, device=rgbs.device)  # one (R,K) draw per pass, reference order
    noisy[..., 3] += n * std
    return noisy, live


class _NoRelease:
    @staticmethod
    def release():
        pass


def _train_eval(net, scene, coarse, rays, z):
    """network forward of one training pass -> (rgbsigma (R,K,4), saved operands).  precision 'f32': the exact, unfused fp32
    chain (validation grade); 'f16x3': the same chain with split-operand (fp32-class) GEMMs on the f16 matrix cores;
    'f16' / 'bf16': the fused kernel's training instantiation (16-bit operand dumps)."""
    ops.saturation_guard_slot(rays.device, 0 if coarse else 1)  # (when the fp16-range guard is armed for this call)
    if net.precision == "f16x3" and FUSED_SPLIT_TRAINING:
        pk = net.packed(coarse, training_pass=True)  # the folded split stream of inference (before tables(): packed() runs the content check)
        return ops.eval_ray_samples_split_train(scene, pk, net.training_tables(coarse, rays, z, scene=scene), rays, z)  # (large grids: only the rows the pass reads)
    if net.precision in ("f32", "f16x3"):
        mlp = net.mlp_coarse if (coarse or net.mlp_fine is None) else net.mlp_fine
        return ops.eval_ray_samples_f32_train(scene, mlp.packed("f32"), rays, z, split=net.precision == "f16x3")
    return ops.eval_ray_samples_train(scene, net.packed(coarse, folded=False, training_pass=True), rays, z)


def _pass_grads(net, mlp, dumps, g_out, scene_NS, want_d_in):
    """-> (grads, d_zlat, d_in, releasable) of one pass at the network's precision"""
    if isinstance(dumps, ops.SplitSaved):
        grads, d_zlat, d_in = ops.mlp_backward_split(mlp.packed("f32"), dumps, g_out, want_d_in=want_d_in)
        return grads, d_zlat, d_in, _NoRelease
    if net.precision in ("f32", "f16x3"):
        grads, d_zlat, d_in = ops.mlp_backward_f32(mlp.packed("f32"), dumps, g_out, want_d_in=want_d_in)
        return grads, d_zlat, d_in, _NoRelease
    return _mlp_grads(None, mlp.packed_bwd(net.precision), dumps, g_out, scene_NS, want_d_in=want_d_in)


def _mlp_grads(mlp_state, packed_bwd, fwd, g_out, scene_NS, want_d_in=False):
    """All parameter gradients of one ResnetFC + d(interpolated latent) [+ d(lin_in operand)] from
    one backward pass.  fwd: ops.TrainDumps of the forward; g_out (P,4) fp32 = dL/d(lin_out output)."""
    # run the 16-bit chain at a power-of-two scale that puts max|g| near 2^6 (exact to undo); the scale is picked
    # on the device (no host sync in the middle of the backward); a non-finite g poisons it with NaN
    sc = ops.grad_scale(g_out)
    bd = ops.mlp_backward(packed_bwd, fwd, g_out, sc[0:1])
    inv_s = sc[1:2]

    prec = packed_bwd.precision
    grads = {}
    # weight gradients of the 512x512 linears: HIP MFMA kernel straight from the 16-bit dumps
    # (fp32 accumulation); the kernel's reduction step writes them in feature order
    jobs, names = [], []
    for b in range(5):
        jobs += [(bd.g_fc0[b], fwd.d_a[b], True, True), (bd.g_fc1[b], fwd.d_n[b], True, True)]
        names += [f"blocks.{b}.fc_0", f"blocks.{b}.fc_1"]
    gzs = [bd.g_x0 if b == 0 else bd.g_fc1[b - 1] for b in range(3)]  # dL/d(residual stream entering block b), per view
    for b in range(3):
        jobs.append((gzs[b], fwd.d_z, True, False))  # latent channels are in natural order
        names.append(f"lin_z.{b}")
    jobs.append((bd.g_x0, fwd.d_in, True, False, 64, 42))  # lin_in: operand (rows,64) = code | viewdir | 0-pad
    names.append("lin_in")
    for name, (dW, db) in zip(names, ops.weight_grad_batched(jobs, prec, 1.0, out_scale_dev=inv_s)):  # one launch, 14 linears
        grads[name + ".weight"], grads[name + ".bias"] = dW, db
    grads["lin_out.weight"], grads["lin_out.bias"] = ops.lin_out_grad(g_out, fwd.d_x5, prec)
    # d z_lat = sum_b dY_b W_z[b] and d(code | viewdir) = dY W_in come out of the same fused chain (pnr_mlp_backward:
    # four more transposed-stream GEMMs on gradient images the kernel already holds), fp32, unscaled
    return grads, bd.d_zlat, (bd.d_in if want_d_in else None), bd


class _RenderFunction(torch.autograd.Function):
    """inputs: cfg (python object), rays (R,8), latent (SB*NS,512,Hl,Wl), 30 coarse params,
    30 fine params (or the coarse ones again when mlp_fine is None).
    outputs: rgb_c, depth_c, weights_c[, rgb_f, depth_f, weights_f]."""

    @staticmethod
    def forward(ctx, cfg, rays, latent, *params):
        # outputs the loss does not use (depths, weights) arrive in backward as None, not as zero tensors torch would
        # have to fill and the compositing backward would have to read
        ctx.set_materialize_grads(False)
        net, noise = cfg["net"], cfg["noise"]
        Kc, Kf, Kfd = cfg["n_coarse"], cfg["n_fine"], cfg["n_fine_depth"]
        scene = net.scene()
        passes = []
        z_c = ops.sample_coarse(rays, noise["u1"], cfg["lindisp"])
        rgbs_c, dumps_c = _train_eval(net, scene, True, rays, z_c)  # the training instantiation keeps the lin_z GEMMs (operands are dumped)
        rgbs_c, live_c = _sigma_noise(rgbs_c, cfg)
        w_c, rgb_c, depth_c = ops.composite(rays, z_c, rgbs_c, cfg["white_bkgd"], want_weights=True)
        passes.append(dict(z=z_c, rgbs=rgbs_c, dumps=dumps_c, coarse=True, live=live_c))
        outs = [rgb_c, depth_c, w_c]
        if Kf > 0:
            n4 = noise.get("n4") if Kfd > 0 else None
            z_f, ranks = ops.sample_fine(rays, w_c, depth_c, z_c, noise.get("u2"), noise.get("u3"), n4,
                                         cfg["depth_std"], cfg["lindisp"], want_ranks=True)
            rgbs_f, dumps_f = _train_eval(net, scene, False, rays, z_f)
            rgbs_f, live_f = _sigma_noise(rgbs_f, cfg)
            w_f, rgb_f, depth_f = ops.composite(rays, z_f, rgbs_f, cfg["white_bkgd"], want_weights=True)
            # depth_c is an OUTPUT of this Function: keeping the tensor itself in ctx would close the cycle
            # output -> grad_fn -> ctx -> output and pin every dump of the step until the cyclic GC runs
            passes.append(dict(z=z_f, rgbs=rgbs_f, dumps=dumps_f, coarse=False, ranks=ranks, n4=n4, depth_c=depth_c.detach(), live=live_f))
            outs += [rgb_f, depth_f, w_f]
        ctx.cfg, ctx.rays, ctx.scene, ctx.passes = cfg, rays, scene, passes
        ctx.latent_shape = latent.shape
        ctx.n_params = len(params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        if ctx.passes is None:
            raise RuntimeError("pixelnerf_amd: backward through this render a second time is not supported -- the operand dumps "
                               "of the forward were handed back after the first backward (retain_graph=True keeps the autograd "
                               "graph, not the 12 KB per point and view of saved operands); sum the losses and call backward once")
        cfg, rays, scene = ctx.cfg, ctx.rays, ctx.scene
        net = cfg["net"]
        dev = rays.device
        need_latent = ctx.needs_input_grad[2]
        # one zeroed grid gradient PER PASS, summed at the end: pnr_latent_scatter's LDS-slab form leaves at most two atomic adds
        # per element and call (its owner workgroups split the points at most two ways on a full-size batch), and two terms onto
        # zero commute -- the latent gradient, and with it the whole step, is then bit-reproducible.  Accumulating the fine and
        # the coarse pass into ONE buffer made the second call's adds land on non-zero values in either order.
        # Large grids take the scatter's tiled form -- one owner workgroup per element, plain read-add-write -- where accumulation is
        # order-free anyway: ONE buffer for all passes (the same bits as the sum of per-pass buffers: 0 + a is exact), which saves a
        # fill and a sum over a grid-sized tensor (DTU: 3 x 184 MB of traffic per step).
        n_buf = len(ctx.passes)
        if need_latent and n_buf > 1 and all(ops.latent_scatter_single_owner(scene, ps["z"].shape[0], ps["z"].shape[1]) for ps in ctx.passes):
            n_buf = 1
        d_lat = torch.zeros((n_buf, ctx.latent_shape[0], ctx.latent_shape[2], ctx.latent_shape[3], ctx.latent_shape[1]),
                            dtype=torch.float32, device=dev) if need_latent else None
        shared = net.mlp_fine is None  # fine pass ran on the coarse network (models.py:242)
        gsum = [None, None]
        extra_depth = None  # dL/d(coarse depth) arriving through the fine pass's depth samples
        for i in reversed(range(len(ctx.passes))):  # fine first: it feeds a depth gradient to coarse
            ps = ctx.passes[i]
            d_rgb, d_depth, d_w = gouts[3 * i], gouts[3 * i + 1], gouts[3 * i + 2]
            R, K = ps["z"].shape
            if d_rgb is None:
                d_rgb = torch.zeros((R, 3), device=dev)
            if ps["coarse"] and extra_depth is not None:
                d_depth = extra_depth if d_depth is None else d_depth + extra_depth
            pos = (not ps["coarse"]) and ps.get("ranks") is not None  # depth samples exist
            cb = ops.composite_backward(rays, ps["z"], ps["rgbs"], cfg["white_bkgd"], d_rgb.contiguous().float(),
                                        None if d_depth is None else d_depth.contiguous().float(),
                                        None if d_w is None else d_w.contiguous().float(), want_dz=pos,
                                        pre_activation=True)  # also through sigmoid / relu (models.py:260-265)
            d_pre, dz = cb if pos else (cb, None)
            if ps.get("live") is not None:  # sigma noise: the compositing kernel saw relu(sigma) + n; relu' of the network's own sigma
                d_pre[..., 3] *= ps["live"]
            g_out = d_pre.reshape(-1, 4)
            mlp = net.mlp_coarse if (ps["coarse"] or shared) else net.mlp_fine
            grads, d_zlat, d_in, bd = _pass_grads(net, mlp, ps["dumps"], g_out, scene.NS, pos)
            slot = 0 if (ps["coarse"] or shared) else 1
            gsum[slot] = grads if gsum[slot] is None else {k: gsum[slot][k] + v for k, v in grads.items()}
            if need_latent:
                ops.latent_scatter(scene, rays, ps["z"], d_zlat, d_lat[i if n_buf > 1 else 0])
            if pos:
                # only the depth samples carry position gradient: compositing part (dz) + network-input part at their
                # sorted positions, through the clamp z = max(min(depth + n*std, far), near)   (nerf.py:157-160,292)
                extra_depth = ops.depth_sample_backward(scene, rays, ps["z"], ps["ranks"], ps["n4"], ps["depth_c"],
                                                        cfg["depth_std"], d_in, d_zlat, dz)
            # every consumer of this pass's dumps is enqueued: the sets go back to the pool (same-stream reuse)
            bd.release()
            ps["dumps"].release()
            ps["dumps"] = None
        ctx.passes = None  # release the 16-bit operand dumps (~12 KB per point and view) as soon as they are used
        out = [None, None, (d_lat[0] if d_lat.shape[0] == 1 else d_lat.sum(0)).permute(0, 3, 1, 2).contiguous() if need_latent else None]
        n_each = len(PARAM_NAMES)
        for slot in range(ctx.n_params // n_each):
            g = gsum[slot]
            out += [None if g is None else g[n] for n in PARAM_NAMES]
        return tuple(out)


class _PointsFunction(torch.autograd.Function):
    """Differentiable PixelNeRFNet.forward on explicit points (src/model/models.py:146-266): (SB*B, 3) points and view
    directions -> (SB*B, 4) = (sigmoid rgb, relu sigma).  The points go through the training kernels as one-sample rays
    (origin = point, direction = viewdir, z = 0).  Gradients: the 30 ResnetFC parameters and the latent grid; the
    points themselves are inputs (no gradient), as on the renderer path."""

    @staticmethod
    def forward(ctx, cfg, xyz, viewdirs, latent, *params):
        ctx.set_materialize_grads(False)
        net, coarse = cfg["net"], cfg["coarse"]
        scene = net.scene()
        R = xyz.shape[0]
        rays = torch.cat([xyz, viewdirs, torch.zeros((R, 2), dtype=torch.float32, device=xyz.device)], dim=1).contiguous()
        z = torch.zeros((R, 1), dtype=torch.float32, device=xyz.device)
        rgbs, dumps = _train_eval(net, scene, coarse, rays, z)
        ctx.cfg, ctx.scene, ctx.rays, ctx.z, ctx.dumps = cfg, scene, rays, z, dumps
        ctx.latent_shape = latent.shape
        out = rgbs.reshape(R, 4)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * (4 + len(PARAM_NAMES))
        if ctx.dumps is None:
            raise RuntimeError("pixelnerf_amd: backward through this forward a second time is not supported (the operand dumps "
                               "were handed back after the first backward); sum the losses and call backward once")
        (out,) = ctx.saved_tensors
        net, coarse = ctx.cfg["net"], ctx.cfg["coarse"]
        g = g.contiguous().float()
        # through the output activations (models.py:260-265): rgb = sigmoid(.), sigma = relu(.)
        d_pre = torch.cat([g[:, :3] * out[:, :3] * (1.0 - out[:, :3]), g[:, 3:] * (out[:, 3:] > 0).float()], dim=1).contiguous()
        mlp = net.mlp_coarse if (coarse or net.mlp_fine is None) else net.mlp_fine
        grads, d_zlat, _, bd = _pass_grads(net, mlp, ctx.dumps, d_pre, ctx.scene.NS, False)
        d_lat = None
        if ctx.needs_input_grad[3]:
            n, c, hl, wl = ctx.latent_shape
            d_lat = torch.zeros((n, hl, wl, c), dtype=torch.float32, device=g.device)
            ops.latent_scatter(ctx.scene, ctx.rays, ctx.z, d_zlat, d_lat)
            d_lat = d_lat.permute(0, 3, 1, 2).contiguous()
        bd.release()
        ctx.dumps.release()
        ctx.dumps = None
        return (None, None, None, d_lat) + tuple(grads[n] for n in PARAM_NAMES)


class _CompositeFunction(torch.autograd.Function):
    """Differentiable alpha compositing around an arbitrary model's per-point (rgb, sigma) (src/render/nerf.py:178-182,
    223-249): forward = pnr_composite, backward = pnr_composite_backward.  Gradients reach the model through `rgbsigma`
    (relu' of the raw sigma is applied inside, as nerf.py:228 applies the relu inside) and the sample positions `z`
    (through the deltas and depth = sum w z) -- the latter is what carries the fine loss back to the coarse depth
    (nerf.py:157-160,292).  Rays are inputs, like on the fused path."""

    @staticmethod
    def forward(ctx, rays, z, rgbsigma, white_bkgd):
        ctx.set_materialize_grads(False)
        w, rgb, depth = ops.composite(rays, z, rgbsigma, white_bkgd, want_weights=True)
        ctx.save_for_backward(rays, z, rgbsigma)
        ctx.white_bkgd = bool(white_bkgd)
        return w, rgb, depth

    @staticmethod
    def backward(ctx, d_w, d_rgb, d_depth):
        rays, z, rgbsigma = ctx.saved_tensors
        if d_w is None and d_rgb is None and d_depth is None:
            return None, None, None, None
        if d_rgb is None:
            d_rgb = torch.zeros((rays.shape[0], 3), dtype=torch.float32, device=rays.device)
        want_dz = ctx.needs_input_grad[1]
        res = ops.composite_backward(rays, z, rgbsigma, ctx.white_bkgd, d_rgb.contiguous().float(),
                                     None if d_depth is None else d_depth.contiguous().float(),
                                     None if d_w is None else d_w.contiguous().float(), want_dz=want_dz, pre_activation=False)
        d_rgbs, dz = res if want_dz else (res, None)
        return None, dz, (d_rgbs if ctx.needs_input_grad[2] else None), None


def composite_autograd(rays, z, rgbsigma, white_bkgd):
    """-> weights (R,K), rgb (R,3), depth (R), differentiable with respect to `rgbsigma` (R,K,4) and `z` (R,K)."""
    return _CompositeFunction.apply(rays, z.contiguous().float(), rgbsigma.contiguous().float(), bool(white_bkgd))


class _SampleFineFunction(torch.autograd.Function):
    """The fine pass's merged sample set (nerf.py:285-295: importance samples from the DETACHED coarse weights, depth samples
    around the coarse depth, sort) with the one gradient the reference has there: z_all -> the depth samples at their sorted
    positions -> through the clamp max(min(depth + n * std, far), near) -> coarse depth (the coarse depth is not detached,
    nerf.py:292).  forward = pnr_sample_fine (which also reports the depth samples' sorted positions)."""

    @staticmethod
    def forward(ctx, rays, weights_c, depth_c, z_coarse, u2, u3, n4, depth_std, lindisp):
        z_all, ranks = ops.sample_fine(rays, weights_c, depth_c, z_coarse, u2, u3, n4, depth_std, lindisp, want_ranks=True)
        ctx.save_for_backward(rays, depth_c, n4, ranks)
        ctx.depth_std = float(depth_std)
        ctx.mark_non_differentiable(ranks)
        return z_all, ranks

    @staticmethod
    def backward(ctx, dz_all, _):
        rays, depth_c, n4, ranks = ctx.saved_tensors
        if dz_all is None:
            return (None,) * 9
        zraw = depth_c.unsqueeze(1) + n4 * ctx.depth_std
        live = (zraw < rays[:, 7:8]) & (zraw > rays[:, 6:7])  # inside the clamp: the gradient passes (nerf.py:160)
        g = torch.gather(dz_all, 1, ranks.long()) * live.to(dz_all.dtype)
        return None, None, g.sum(dim=1), None, None, None, None, None, None


def sample_fine_autograd(rays, weights_c, depth_c, z_coarse, u2, u3, n4, depth_std, lindisp):
    """ops.sample_fine, differentiable with respect to the coarse depth (only the n_fine_depth samples depend on it)."""
    return _SampleFineFunction.apply(rays, weights_c, depth_c, z_coarse, u2, u3, n4, float(depth_std), bool(lindisp))[0]


class _LinearFunction(torch.autograd.Function):
    """y = [residual +] [relu](x) W^T + b as ONE autograd node around pnr_linear / pnr_linear_backward: the building block of
    ResnetFCs of non-shipped shapes (model/resnetfc.py `_forward_composed`; src/model/resnetfc.py:53-62,147,175-183).  Saves x
    (the pre-ReLU input: relu' = [x > 0] is applied inside the data-gradient kernel) -- nothing else."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, relu_in, precision):
        ctx.save_for_backward(x, weight)
        ctx.relu_in, ctx.precision, ctx.has_bias = bool(relu_in), precision, bias is not None
        return ops.linear(x, weight, bias, relu_in=relu_in, residual=residual, precision=precision)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        need_dx, need_dw, need_db, need_res = ctx.needs_input_grad[:4]
        dy = dy.contiguous().float()
        dx = dw = db = None
        if need_dx or need_dw or (need_db and ctx.has_bias):
            dx, dw, db = ops.linear_backward(dy, x, weight, relu_in=ctx.relu_in, need_dx=need_dx, need_dw=need_dw,
                                             need_db=need_db and ctx.has_bias, precision=ctx.precision)
        return dx, (dw if need_dw else None), db, (dy if need_res else None), None, None


def linear_autograd(x, weight, bias=None, relu_in=False, residual=None, precision="f16x3"):
    """Differentiable `ops.linear` (plain kernel call when nothing requires grad)."""
    x = x.float()
    if not x.is_contiguous():
        x = x.contiguous()
    if residual is not None:
        residual = residual.float().contiguous()
    return _LinearFunction.apply(x, weight, bias, residual, bool(relu_in), precision)


def points_autograd(net, xyz, viewdirs, coarse):
    """net(xyz, viewdirs) with autograd: (SB,B,3) x 2 -> (SB,B,4)."""
    if xyz.requires_grad or viewdirs.requires_grad:
        raise NotImplementedError("gradients with respect to the query points / view directions are not implemented "
                                  "(the renderer path does not need them: rays are inputs)")
    SB, B, _ = xyz.shape
    latent = net.encoder.latent
    if net.stop_encoder_grad:
        latent = latent.detach()
    mlp = net.mlp_coarse if (coarse or net.mlp_fine is None) else net.mlp_fine
    out = _PointsFunction.apply(dict(net=net, coarse=coarse), xyz.reshape(-1, 3).float().contiguous(),
                                viewdirs.reshape(-1, 3).float().contiguous(), latent, *mlp.ordered_params(PARAM_NAMES))
    return out.reshape(SB, B, 4)


def render_autograd(renderer, net, rays, noise, want_weights):
    """Differentiable twin of the one-call inference path; returns {coarse:{...}, fine:{...}} of
    flat tensors like ops.render_forward."""
    Kf = renderer.n_fine if renderer.using_fine else 0
    cfg = dict(net=net, noise=noise, n_coarse=renderer.n_coarse, n_fine=Kf,
               n_fine_depth=min(renderer.n_fine_depth, Kf), depth_std=renderer.depth_std,
               noise_std=float(renderer.noise_std) if renderer.training else 0.0,
               white_bkgd=bool(renderer.white_bkgd), lindisp=bool(renderer.lindisp),
               ray_id_offset=int(getattr(renderer, "ray_id_offset", 0)), ray_id_stride=int(getattr(renderer, "ray_id_stride", 0)),
               num_objs=int(net.num_objs))
    latent = net.encoder.latent
    if net.stop_encoder_grad:
        latent = latent.detach()
    mlps = [net.mlp_coarse] + ([net.mlp_fine] if net.mlp_fine is not None else [])
    params = []
    for m in mlps:
        params += m.ordered_params(PARAM_NAMES)
    sync = getattr(net, "_grad_sync", None)
    if sync is not None:  # multi-process training (dist.ShardedRenderWrapper): identity here, ONE gradient all-reduce in backward
        latent, params = sync(latent, params)
    outs = _RenderFunction.apply(cfg, rays, latent, *params)
    res = {"coarse": {"rgb": outs[0], "depth": outs[1], "weights": outs[2]}}
    if Kf > 0:
        res["fine"] = {"rgb": outs[3], "depth": outs[4], "weights": outs[5]}
    if not want_weights:
        for v in res.values():
            v.pop("weights")
    return res
