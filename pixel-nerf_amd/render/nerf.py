"""
NeRFRenderer / _RenderWrapper with the reference's API (src/render/nerf.py:15-371); the work
is done by libpixelnerf_hip.so.

With a pixelnerf_amd PixelNeRFNet the whole forward (coarse sampling -> fused network ->
compositing -> inverse-CDF + depth resampling + sort -> fused network -> compositing) is one
C call (pnr_render_forward).  Any other `model(xyz, coarse=, viewdirs=)` callable still works:
sampling and compositing run as HIP kernels around the caller's model, chunked by
eval_batch_size exactly like the reference -- and differentiably (autograd.composite_autograd /
sample_fine_autograd), so a renderer around an arbitrary nn.Module trains as it does in the reference.

Random numbers.  `rng="philox"` (default in eval mode with a pixelnerf_amd net): the sampling kernels draw from a
counter-based generator (Philox4x32-10) keyed by a 64-bit seed -- `torch.initial_seed()` of the ray device's generator
mixed with that generator's Philox offset (advanced per call, host side), so `torch.manual_seed(s)` makes a run
reproducible exactly as it does for the reference, without any device-side launch or HBM traffic for noise; a draw depends only on (seed, global ray id, draw, index), so chunked or multi-GPU
sharded renders give the same image as one call.  `rng="torch"` (and training, and generic model callables): torch's
generator on the ray device, in the reference's draw order (nerf.py:111,135,141,158).  Tests inject pre-drawn noise
through `_noise`.
"""
import torch

from .. import ops
from ..util.dotmap import DotMap


class _RenderWrapper(torch.nn.Module):
    """src/render/nerf.py:15-42."""

    def __init__(self, net, renderer, simple_output):
        super().__init__()
        self.net = net
        self.renderer = renderer
        self.simple_output = simple_output

    def forward(self, rays, want_weights=False):
        if rays.shape[0] == 0:
            return (torch.zeros(0, 3, device=rays.device), torch.zeros(0, device=rays.device))
        with torch.profiler.record_function("render_par"):
            outputs = self.renderer(self.net, rays, want_weights=want_weights and not self.simple_output)
        if self.simple_output:
            if self.renderer.using_fine:
                return outputs.fine.rgb, outputs.fine.depth
            return outputs.coarse.rgb, outputs.coarse.depth
        return outputs.toDict()


class NeRFRenderer(torch.nn.Module):
    """NeRF renderer; parameters as src/render/nerf.py:45-96."""

    def __init__(self, n_coarse=128, n_fine=0, n_fine_depth=0, noise_std=0.0, depth_std=0.01,
                 eval_batch_size=100000, white_bkgd=False, lindisp=False, sched=None, rng="philox"):
        super().__init__()
        if rng not in ("philox", "torch"):
            raise ValueError("rng must be 'philox' (in-kernel counter-based draws) or 'torch'")
        self.rng = rng
        self._seed_override = None  # set by the multi-device wrapper: every shard of one call uses the same key
        self.ray_id_offset = 0  # placement of this call's rays inside a larger ray set (set by sharding wrappers)
        self.ray_id_stride = 0
        self.n_coarse, self.n_fine, self.n_fine_depth = n_coarse, n_fine, n_fine_depth
        self.noise_std, self.depth_std = noise_std, depth_std
        self.eval_batch_size = eval_batch_size
        self.white_bkgd = white_bkgd
        self.lindisp = lindisp
        if lindisp:
            print("Using linear displacement rays")
        self.using_fine = n_fine > 0
        self.sched = sched
        if sched is not None and len(sched) == 0:
            self.sched = None
        self.register_buffer("iter_idx", torch.tensor(0, dtype=torch.long), persistent=True)
        self.register_buffer("last_sched", torch.tensor(0, dtype=torch.long), persistent=True)

    # ---- stage methods (same names / shapes as the reference; HIP kernels underneath) ----
    def sample_coarse(self, rays, _u=None):
        """nerf.py:98-118.  rays (B,8) -> (B,Kc)."""
        u = torch.rand(rays.shape[0], self.n_coarse, device=rays.device) if _u is None else _u
        return ops.sample_coarse(rays, u, self.lindisp)

    def sample_fine(self, rays, weights, _u2=None, _u3=None):
        """nerf.py:120-148.  -> (B, Kf-Kfd), unsorted, like the reference."""
        n = self.n_fine - self.n_fine_depth
        B, dev = rays.shape[0], rays.device
        u2 = torch.rand(B, n, dtype=torch.float32, device=dev) if _u2 is None else _u2
        u3 = torch.rand(B, n, dtype=torch.float32, device=dev) if _u3 is None else _u3
        # the kernel returns sorted([z_coarse, z_fine]); feeding `near` as every coarse sample and
        # dropping the Kc smallest leaves the sorted importance samples (their order is irrelevant
        # downstream: forward() sorts everything, :295)
        Kc = weights.shape[1]
        z0 = rays[:, 6:7].expand(-1, Kc).contiguous()
        z = ops.sample_fine(rays, weights.detach(), None, z0, u2, u3, None, self.depth_std, self.lindisp)
        return z[:, Kc:].contiguous()

    def sample_fine_depth(self, rays, depth, _n=None):
        """nerf.py:150-161.  -> (B, Kfd)."""
        n = torch.randn(rays.shape[0], self.n_fine_depth, device=rays.device) if _n is None else _n
        z0 = rays[:, 6:7].contiguous()
        z = ops.sample_fine(rays, None, depth, z0, None, None, n, self.depth_std, self.lindisp)
        return z[:, 1:].contiguous()  # drop the placeholder coarse sample (= near, the minimum)

    def composite(self, model, rays, z_samp, coarse=True, sb=0):
        """nerf.py:163-249 for an arbitrary model callable: points/viewdirs, chunked model
        calls (eval_batch_size), then the HIP compositing kernel.
        :return weights (B,K), rgb (B,3), depth (B)"""
        with torch.profiler.record_function("renderer_composite"):  # nerf.py:175
            return self._composite(model, rays, z_samp, coarse, sb)

    def _composite(self, model, rays, z_samp, coarse, sb):
        B, K = z_samp.shape
        points = (rays[:, None, :3] + z_samp.unsqueeze(2) * rays[:, None, 3:6]).reshape(-1, 3)
        use_viewdirs = hasattr(model, "use_viewdirs") and model.use_viewdirs
        if sb > 0:
            points = points.reshape(sb, -1, 3)
            eval_batch_size = (self.eval_batch_size - 1) // sb + 1
            dim = 1
        else:
            eval_batch_size = self.eval_batch_size
            dim = 0
        val_all = []
        split_points = torch.split(points, eval_batch_size, dim=dim)
        if use_viewdirs:
            viewdirs = rays[:, None, 3:6].expand(-1, K, -1)
            viewdirs = viewdirs.reshape(sb, -1, 3) if sb > 0 else viewdirs.reshape(-1, 3)
            for pnts, dirs in zip(split_points, torch.split(viewdirs, eval_batch_size, dim=dim)):
                val_all.append(model(pnts.contiguous(), coarse=coarse, viewdirs=dirs.contiguous()))
        else:
            for pnts in split_points:
                val_all.append(model(pnts.contiguous(), coarse=coarse))
        out = torch.cat(val_all, dim=dim).reshape(B, K, -1)
        if self.training and self.noise_std > 0.0:
            out = torch.cat([out[..., :3], out[..., 3:4] + torch.randn_like(out[..., 3:4]) * self.noise_std], -1)
        rgbs = out[..., :4].contiguous()
        if torch.is_grad_enabled() and (rgbs.requires_grad or z_samp.requires_grad):
            # training with an arbitrary model: the compositing kernels as an autograd node (pnr_composite /
            # pnr_composite_backward); gradients reach the model through its outputs and, for the depth samples of the fine
            # pass, through the sample positions (points above are torch ops on z_samp)
            from ..autograd import composite_autograd
            return composite_autograd(rays, z_samp, rgbs, self.white_bkgd)
        return ops.composite(rays, z_samp, rgbs, self.white_bkgd, want_weights=True)

    # ---- forward ----
    def _draw_noise(self, R, dev):
        """torch draws in the reference's order: rand_like (R,Kc) :111, rand (R,Kf-Kfd) :135,
        rand_like :141, randn_like (R,Kfd) :158."""
        noise = {"u1": torch.rand(R, self.n_coarse, device=dev)}
        if self.using_fine:
            n_imp = self.n_fine - self.n_fine_depth
            if n_imp > 0:
                noise["u2"] = torch.rand(R, n_imp, dtype=torch.float32, device=dev)
                noise["u3"] = torch.rand(R, n_imp, dtype=torch.float32, device=dev)
            if self.n_fine_depth > 0:
                noise["n4"] = torch.randn(R, self.n_fine_depth, device=dev)
        return noise

    def forward(self, model, rays, want_weights=False, _noise=None):
        """src/render/nerf.py:251-303.
        :param model nerf model: (SB,B,3) points [+ viewdirs] -> (SB,B,4) rgb sigma
        :param rays [origins(3), directions(3), near, far] (SB,B,8)
        :return DotMap {coarse:{rgb (SB,B,3), depth (SB,B)[, weights (SB,B,K)]}, fine:{...}}"""
        with torch.profiler.record_function("renderer_forward"):  # the reference's scope name (nerf.py:264)
            return self._forward(model, rays, want_weights, _noise)

    def _forward(self, model, rays, want_weights, _noise):
        if self.sched is not None and self.last_sched.item() > 0:
            self.n_coarse = self.sched[1][self.last_sched.item() - 1]
            self.n_fine = self.sched[2][self.last_sched.item() - 1]
        assert len(rays.shape) == 3
        SB = rays.shape[0]
        rays = rays.reshape(-1, 8).float().contiguous()
        R = rays.shape[0]
        # pixelnerf_amd.PixelNeRFNet in the shipped configuration: the fused kernels.  Any other model -- including a PixelNeRFNet
        # configured outside what the fused kernels implement (composed forward) -- is a callable to the reference's control flow.
        fused = hasattr(model, "scene") and hasattr(model, "packed") and (not hasattr(model, "fused_supported") or model.fused_supported())
        # (inside a HIP-graph capture the generator's offset cannot be read on the host: torch's own graph-safe draws are used)
        seeded = (_noise is None and fused and self.rng == "philox" and not (self.training and torch.is_grad_enabled())
                  and not (rays.is_cuda and torch.cuda.is_current_stream_capturing()))
        noise = _noise if (_noise is not None or seeded) else self._draw_noise(R, rays.device)
        Kf = self.n_fine if self.using_fine else 0
        Kfd = min(self.n_fine_depth, Kf)

        if fused:  # pixelnerf_amd.PixelNeRFNet: one C call
            model._check_supported()
            needs_grad = torch.is_grad_enabled() and (
                model.mlp_coarse.any_requires_grad()
                or (model.mlp_fine is not None and model.mlp_fine.any_requires_grad())
                or (model.encoder.latent.requires_grad and not model.stop_encoder_grad))
            if self.training and self.noise_std > 0.0 and not needs_grad:
                raise NotImplementedError("noise_std > 0 in train mode is implemented on the differentiable path (grad enabled); "
                                          "the reference adds the noise only while training (nerf.py:225-226)")
            if needs_grad:  # training: differentiable path (HIP forward with operand dumps + HIP backward)
                from ..autograd import render_autograd
                model._check_trainable()
                if noise is None:
                    noise = self._draw_noise(R, rays.device)
                guarded = model._guard_begin(training=True)
                try:
                    res = render_autograd(self, model, rays, noise, want_weights)
                finally:
                    if guarded:
                        model._guard_end()
            else:
                # mlp_fine is None (eval/eval.py:140): pass no fine network, so the fine pass re-uses the coarse pass's
                # outputs at the shared sample positions instead of evaluating them again
                own_fine = Kf > 0 and getattr(model, "mlp_fine", None) is not None
                pk_c, pk_f = model.packed(True), (model.packed(False) if own_fine else None)  # before tables(): see PixelNeRFNet.tables
                guarded = model._guard_begin()  # fp16-range guard of the fp32-class kernels: first call on new weights / scene
                try:
                    tc = model.tables(True)   # (a fold that happens now is guarded too: grid values / lin_z weights)
                    if guarded:
                        ops.saturation_guard_slot(rays.device, 1)
                    tf = model.tables(False) if (own_fine and tc is not None) else None
                    if guarded:
                        ops.saturation_guard_slot(rays.device, 0)
                    res = ops.render_forward(model.scene(), pk_c, pk_f,
                                             rays, self.n_coarse, Kf, Kfd, noise, depth_std=self.depth_std,
                                             white_bkgd=self.white_bkgd, lindisp=self.lindisp, want_weights=want_weights,
                                             tables=None if tc is None else (tc, tf),
                                             seed=self._next_seed(rays.device) if seeded else 0,
                                             ray_id_offset=self.ray_id_offset, ray_id_stride=self.ray_id_stride)
                finally:
                    if guarded:
                        model._guard_end()
            outputs = DotMap(coarse=self._format(res["coarse"], SB, want_weights))
            if Kf > 0:
                outputs.fine = self._format(res["fine"], SB, want_weights)
            return outputs

        # generic model callable: reference control flow, HIP kernels for every renderer stage
        z_coarse = self.sample_coarse(rays, _u=noise["u1"])
        wc, rgbc, depthc = self.composite(model, rays, z_coarse, coarse=True, sb=SB)
        outputs = DotMap(coarse=self._format(dict(rgb=rgbc, depth=depthc, weights=wc), SB, want_weights))
        if Kf > 0:
            n4 = noise.get("n4") if Kfd > 0 else None
            if n4 is not None and torch.is_grad_enabled() and depthc.requires_grad:
                # nerf.py:292: the coarse depth is not detached -- the depth samples carry the fine loss back to it
                from ..autograd import sample_fine_autograd
                z_all = sample_fine_autograd(rays, wc.detach(), depthc, z_coarse, noise.get("u2"), noise.get("u3"), n4,
                                             self.depth_std, self.lindisp)
            else:
                z_all = ops.sample_fine(rays, wc.detach(), depthc, z_coarse, noise.get("u2"), noise.get("u3"), n4,
                                        self.depth_std, self.lindisp)
            wf, rgbf, depthf = self.composite(model, rays, z_all, coarse=False, sb=SB)
            outputs.fine = self._format(dict(rgb=rgbf, depth=depthf, weights=wf), SB, want_weights)
        return outputs

    def _next_seed(self, device):
        """64-bit Philox key of this call, from the ray device's torch generator: (seed, Philox offset) mixed by a
        splitmix64 finaliser, and the generator's offset is advanced as a torch.rand launch would advance it.  Host
        arithmetic only (no launch, no sync); `torch.manual_seed(s)` resets it exactly like it resets the reference's
        draws, successive calls get fresh draws."""
        if self._seed_override is not None:
            return self._seed_override
        gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
        base, off = gen.initial_seed(), gen.get_offset()
        gen.set_offset(off + 4)
        x = (base + 0x9E3779B97F4A7C15 * (off // 4 + 1)) & (2 ** 64 - 1)
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & (2 ** 64 - 1)
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & (2 ** 64 - 1)
        return x ^ (x >> 31)

    @staticmethod
    def _format(d, SB, want_weights):
        """nerf.py:305-316."""
        ret = DotMap(rgb=d["rgb"].reshape(SB, -1, 3), depth=d["depth"].reshape(SB, -1))
        if want_weights:
            ret.weights = d["weights"].reshape(SB, -1, d["weights"].shape[-1])
        return ret

    def sched_step(self, steps=1):
        """nerf.py:318-338."""
        if self.sched is None:
            return
        self.iter_idx += steps
        while (self.last_sched.item() < len(self.sched[0])
               and self.iter_idx.item() >= self.sched[0][self.last_sched.item()]):
            self.n_coarse = self.sched[1][self.last_sched.item()]
            self.n_fine = self.sched[2][self.last_sched.item()]
            print("INFO: NeRF sampling resolution changed on schedule ==> c", self.n_coarse, "f", self.n_fine)
            self.last_sched += 1

    @classmethod
    def from_conf(cls, conf, white_bkgd=False, lindisp=False, eval_batch_size=100000):
        """nerf.py:340-352."""
        return cls(conf.get_int("n_coarse", 128), conf.get_int("n_fine", 0),
                   n_fine_depth=conf.get_int("n_fine_depth", 0), noise_std=conf.get_float("noise_std", 0.0),
                   depth_std=conf.get_float("depth_std", 0.01), white_bkgd=conf.get_float("white_bkgd", white_bkgd),
                   lindisp=lindisp, eval_batch_size=conf.get_int("eval_batch_size", eval_batch_size),
                   sched=conf.get_list("sched", None), rng=conf.get_string("rng", "philox"))

    def bind_parallel(self, net, gpus=None, simple_output=False):
        """nerf.py:354-371.  Returns a module callable as `(rays (SB,B,8), want_weights=False)`.
        The reference wraps it in single-process torch.nn.DataParallel(dim=1), which re-broadcasts
        the whole network and feature grid on every call.  Here multi-GPU means one process per
        GPU (torchrun) over RCCL: when torch.distributed is initialised with world_size > 1 and
        more than one GPU is requested, rays are sharded on dim 1 across ranks and the results
        all-gathered (pixelnerf_amd.dist.ShardedRenderWrapper)."""
        wrapped = _RenderWrapper(net, self, simple_output=simple_output)
        if gpus is not None and len(gpus) > 1:
            import torch.distributed as dist
            print("Using multi-GPU", gpus)
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                from ..dist import ShardedRenderWrapper
                return ShardedRenderWrapper(wrapped)
            # single process, several devices -- what `eval/eval.py --gpu_id "0 1"` asks for (nerf.py:367-371)
            return _MultiDeviceRenderWrapper(net, self, simple_output, gpus)
        return wrapped


class _MultiDeviceRenderWrapper(torch.nn.Module):
    """Single-process counterpart of the reference's DataParallel(_RenderWrapper, gpus, dim=1) (nerf.py:367-371): rays
    are split on dim 1 across the listed devices, every device renders its slice with its own replica of the
    network (weight VALUES copied when they change; the encoded scene -- feature grid, poses, focal, c -- is copied per
    call, as DataParallel's replicate does), results are concatenated on the first device.  Kernel launches are
    asynchronous, so the slices run concurrently.
    Training works as it does through DataParallel (train/train.py:75): under grad, the autograd inputs of a replica's
    render are the SOURCE network's parameters and `encoder.latent` moved to the replica's device with the
    differentiable `.to()`, so every shard's gradient flows back across devices and accumulates in the source
    parameters' `.grad` -- the caller's optimizer steps the source network, the replicas are refreshed on the next call.
    (Across processes: pixelnerf_amd.dist.ShardedRenderWrapper, one all_reduce per step.)"""

    def __init__(self, net, renderer, simple_output, gpus):
        super().__init__()
        self.net, self.renderer, self.simple_output = net, renderer, simple_output
        self.devices = [torch.device("cuda", int(g)) for g in gpus]
        self._replicas = {}  # device index in the list -> [weights fingerprint, net replica, renderer replica]

    def _weights_key(self):
        src = self.net
        mlps = [m for m in (src.mlp_coarse, src.mlp_fine) if m is not None]
        return tuple(m._fingerprint() for m in mlps) + (src.mlp_fine is None,)

    def _replica(self, i):
        import copy
        fp = self._weights_key()
        hit = self._replicas.get(i)
        if hit is None:
            # the copy carries parameters and buffers only: PixelNeRFNet / ResnetFC / SpatialEncoder drop their per-process caches
            # (scene descriptor, folded tables, packed streams, encoder HIP graphs) in __getstate__; the feature grid is not
            # dragged through the copy either -- the replica gets this call's grid below
            src = self.net
            grid, nhwc = src.encoder.latent, getattr(src.encoder, "_nhwc", None)
            src.encoder.latent = torch.empty(0)
            try:
                rep = copy.deepcopy(src).to(self.devices[i])
            finally:
                src.encoder.latent, src.encoder._nhwc = grid, nhwc
            for p in rep.parameters():
                p.requires_grad_(False)  # gradients go to the SOURCE parameters (see forward)
            hit = [fp, rep, copy.deepcopy(self.renderer).to(self.devices[i])]
            self._replicas[i] = hit
        elif hit[0] != fp:
            # the source weights changed (optimizer step, load_state_dict, ...): refresh the VALUES in place, re-pack lazily
            with torch.no_grad():
                for rp, sp in zip(hit[1].parameters(), self.net.parameters()):
                    rp.copy_(sp, non_blocking=True)
            for m in (hit[1].mlp_coarse, hit[1].mlp_fine):
                if m is not None:
                    m.invalidate_packed()
            hit[0] = fp
        _, rep, rend = hit
        dev = self.devices[i]
        src = self.net
        rep.encoder.latent = src.encoder.latent.detach().to(dev, non_blocking=True)
        rep.encoder.latent_scaling = src.encoder.latent_scaling.to(dev, non_blocking=True)
        rep.poses, rep.focal, rep.c = src.poses.to(dev, non_blocking=True), src.focal.to(dev, non_blocking=True), src.c.to(dev, non_blocking=True)
        rep.image_shape = src.image_shape.to(dev, non_blocking=True)
        rep.num_objs, rep.num_views_per_obj, rep.mlp_fine = src.num_objs, src.num_views_per_obj, (rep.mlp_fine if src.mlp_fine is not None else None)
        rep.precision, rep.fold, rep.stop_encoder_grad = src.precision, src.fold, src.stop_encoder_grad
        for k in ("n_coarse", "n_fine", "n_fine_depth", "using_fine", "white_bkgd", "lindisp", "depth_std", "noise_std"):
            setattr(rend, k, getattr(self.renderer, k))
        rend.train(self.renderer.training)
        rep.train(src.training)
        return rep, rend

    def forward(self, rays, want_weights=False):
        from ..autograd import PARAM_NAMES
        from ..dist import shard_bounds
        src = self.net
        lat = src.encoder.latent
        training = torch.is_grad_enabled() and (any(p.requires_grad for p in src.parameters())
                                                or (torch.is_tensor(lat) and lat.requires_grad and not src.stop_encoder_grad))
        B, n = rays.shape[1], len(self.devices)
        seed = self.renderer._next_seed(self.devices[0])  # one key per call, shared by all shards
        outs = []
        for i, dev in enumerate(self.devices):
            lo, hi = shard_bounds(B, i, n)
            if hi == lo:
                continue
            rep, rend = self._replica(i)
            rend.ray_id_offset, rend.ray_id_stride, rend._seed_override = lo, B, seed  # same draws as one device
            if training:
                # autograd inputs of this shard = the source tensors, moved differentiably: the shard's gradients arrive in
                # the source parameters' .grad (summed over the shards by autograd's accumulation)
                mlps = [m for m in (src.mlp_coarse, src.mlp_fine) if m is not None]
                src_params = [p for m in mlps for p in m.ordered_params(PARAM_NAMES)]
                rep.encoder.latent.requires_grad_(lat.requires_grad)  # the replica's renderer takes the differentiable path
                for rp, sp in zip([p for m in (rep.mlp_coarse, rep.mlp_fine) if m is not None for p in m.ordered_params(PARAM_NAMES)], src_params):
                    rp.requires_grad_(sp.requires_grad)
                rep._grad_sync = (lambda latent, params, _d=dev, _sp=src_params:
                                  ((lat.to(_d) if lat.requires_grad and not src.stop_encoder_grad else latent), [p.to(_d) for p in _sp]))
            try:
                with torch.cuda.device(dev):
                    part = _RenderWrapper(rep, rend, self.simple_output)(rays[:, lo:hi].to(dev, non_blocking=True), want_weights=want_weights)
            finally:
                if training:
                    rep._grad_sync = None
            outs.append(part)
        home = self.devices[0]
        if self.simple_output:
            return tuple(torch.cat([o[j].to(home) for o in outs], dim=1) for j in range(2))
        return {k: {kk: torch.cat([o[k][kk].to(home) for o in outs], dim=1) for kk in outs[0][k]} for k in outs[0]}
