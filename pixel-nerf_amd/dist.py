"""
Multi-GPU rendering: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference's only parallelism is single-process torch.nn.DataParallel(dim=1)
(src/render/nerf.py:367-371), which on EVERY call re-broadcasts the whole wrapper (ResNet-34,
both MLPs and the encoded feature grid), scatters the rays and gathers the outputs.  Rays are
independent units, so here:

  * weights live on every rank (loaded once);
  * after `net.encode()` on one rank, `broadcast_encoded()` ships the feature grid with ONE
    collective (camera metadata rides in front of the grid in the same buffer): sn64 2 MiB, srn 16 MiB, DTU 176 MiB;
  * each rank renders a contiguous slice of the rays on dim 1 (`shard_bounds`), exactly the
    split DataParallel(dim=1) makes;
  * results come back with ONE all_gather of the packed outputs, (rgb | depth) = 16 B/ray (`ShardedRenderWrapper`).

No collective sits on the per-sample data path.  The sharding helpers are pure index
arithmetic and are covered by world_size-2 gloo tests on CPU (tests/test_dist_gloo.py).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous near-equal split of n items: first (n % world) ranks get one extra."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_HDR = 8  # int64 header words, carried as exact small floats in front of the metadata


def _flat_fanout(buf, src, group):
    """`buf` from rank src to every other rank as ONE group of point-to-point transfers (RCCL send/recv fused by
    batch_isend_irecv): on MI355X's fully connected xGMI every receiver has its own link to the source, so the
    7 transfers of an 8-GPU node run concurrently at link rate and each link carries the buffer exactly once
     -- unlike a ring/tree broadcast that relays it.  Which form wins is a measurement
    (`bench.py --bcast flat|tree`)."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if rank == src:
        ops = [dist.P2POp(dist.isend, buf, dist.get_global_rank(group, r) if group is not None else r, group=group)
               for r in range(world) if r != src]
    else:
        ops = [dist.P2POp(dist.irecv, buf, dist.get_global_rank(group, src) if group is not None else src, group=group)]
    for w in (dist.batch_isend_irecv(ops) if ops else []):
        w.wait()


def broadcast_encoded(net, src=0, group=None, latent_shape=None, algo="tree", layout="nchw"):
    """Make rank `src`'s encode() state current on every rank with ONE collective when the receivers know the
    grid shape (`latent_shape=(NV,512,Hl,Wl)`, the normal case: every rank knows the dataset's image size and view
    count): a single flat broadcast of [header | poses, focal, c, image_shape, latent_scaling | feature grid].
    Without `latent_shape` a first 8-word broadcast announces the shapes (two collectives in total).  No host
    synchronisation on the sending side; receivers read the header back only in the shape-discovery form.
    algo: "tree" = dist.broadcast (RCCL picks ring / tree), "flat" = one point-to-point transfer per receiver
    (_flat_fanout).
    layout: "nchw" ships `encoder.latent` as the reference holds it; "nhwc" ships the channel-last copy the fused kernels read
    (`encoder.latent_nhwc()`, which the source already has after an inference encode): the receivers install it as the
    cached channel-last grid and expose `encoder.latent` as its (N,C,H,W)-shaped permuted VIEW -- same values for
    `SpatialEncoder.index`, and no per-rank 176 MiB NCHW -> NHWC transpose on the DTU grid after every broadcast."""
    if algo not in ("tree", "flat"):
        raise ValueError("broadcast_encoded: algo must be 'tree' or 'flat'")
    if layout not in ("nchw", "nhwc"):
        raise ValueError("broadcast_encoded: layout must be 'nchw' or 'nhwc'")
    dev = net.poses.device
    is_src = dist.get_rank(group) == src
    if is_src:
        lat = net.encoder.latent
        shape = tuple(lat.shape)
        hdr = torch.tensor([shape[0], shape[1], shape[2], shape[3], net.num_views_per_obj, net.num_objs,
                            net.focal.reshape(-1, 2).shape[0], net.c.reshape(-1, 2).shape[0]], dtype=torch.float32, device=dev)
    elif latent_shape is None:
        hdr = torch.zeros(_HDR, dtype=torch.float32, device=dev)
    if latent_shape is None:
        dist.broadcast(hdr, src, group=group)
        NV, C, Hl, Wl, NS, SB, nf, nc = [int(v) for v in hdr.tolist()]
    else:
        NV, C, Hl, Wl = [int(v) for v in latent_shape]
        NS = SB = nf = nc = None
    # metadata block has a fixed upper bound so that its size does not depend on values only the source knows:
    # poses NV*12, focal <= NV*2, c <= NV*2, image_shape 2, latent_scaling 2
    n_meta = _HDR + NV * 12 + NV * 2 + NV * 2 + 4
    n_lat = NV * C * Hl * Wl
    buf = torch.empty(n_meta + n_lat, dtype=torch.float32, device=dev)
    if is_src:
        if shape != (NV, C, Hl, Wl):
            raise ValueError(f"broadcast_encoded: latent_shape {latent_shape} does not match the encoded grid {shape}")
        meta = torch.zeros(n_meta, dtype=torch.float32, device=dev)
        meta[:_HDR] = hdr
        parts = [net.poses.reshape(-1).float(), net.focal.reshape(-1).float(), net.c.reshape(-1).float(),
                 net.image_shape.reshape(-1).float(), net.encoder.latent_scaling.reshape(-1).float()]
        offs = [_HDR, _HDR + NV * 12, _HDR + NV * 14, _HDR + NV * 16, _HDR + NV * 16 + 2]
        for o, t in zip(offs, parts):
            meta[o:o + t.numel()] = t.to(dev)
        buf[:n_meta] = meta
        buf[n_meta:] = (net.encoder.latent_nhwc() if layout == "nhwc" else net.encoder.latent).reshape(-1)
    if algo == "flat":
        _flat_fanout(buf, src, group)          # THE feature-grid transfer as a 1 -> (N-1) fan-out
    else:
        dist.broadcast(buf, src, group=group)  # THE feature-grid broadcast (metadata rides in front of it)
    if not is_src:
        h = buf[:_HDR].tolist()
        NS, SB, nf, nc = int(h[4]), int(h[5]), int(h[6]), int(h[7])
        if layout == "nhwc":
            nhwc = buf[n_meta:].reshape(NV, Hl, Wl, C)
            lat = nhwc.permute(0, 3, 1, 2)  # (NV,C,Hl,Wl) view of the channel-last buffer
            net.encoder.latent = lat
            net.encoder._nhwc = ((lat.data_ptr(), lat._version, tuple(lat.shape)), nhwc)  # latent_nhwc() cache: no transpose
        else:
            net.encoder.latent = buf[n_meta:].reshape(NV, C, Hl, Wl)
        net.poses = buf[_HDR:_HDR + NV * 12].reshape(NV, 3, 4).clone()
        net.focal = buf[_HDR + NV * 12:_HDR + NV * 12 + nf * 2].reshape(nf, 2).clone()
        net.c = buf[_HDR + NV * 14:_HDR + NV * 14 + nc * 2].reshape(nc, 2).clone()
        net.image_shape = buf[_HDR + NV * 16:_HDR + NV * 16 + 2].clone()
        net.encoder.latent_scaling = buf[_HDR + NV * 16 + 2:_HDR + NV * 16 + 4].clone()
        net.num_views_per_obj, net.num_objs = NS, SB
    return net


def _gather_dim1(t, sizes, group):
    """all_gather tensors that differ in size along dim 1 (pad to the largest shard, trim)."""
    world = len(sizes)
    m = max(sizes)
    if t.shape[1] < m:
        pad = torch.zeros((t.shape[0], m - t.shape[1]) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
        t = torch.cat([t, pad], dim=1)
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t.contiguous(), group=group)
    return torch.cat([b[:, :s] for b, s in zip(bufs, sizes)], dim=1)


class _BucketAllReduce(torch.autograd.Function):
    """Identity on a list of tensors whose backward sums their gradients over the ranks with ONE all_reduce of one
    flattened bucket (RCCL ring / tree over xGMI: one large message instead of 61 small ones).  Autograd calls the
    backward of a multi-output Function once, when the gradients of all its outputs are known -- i.e. at the end of
    the step's backward through the renderer -- so the reduction needs no hook and no explicit call by the trainer."""

    @staticmethod
    def forward(ctx, group, stats, *tensors):
        ctx.group, ctx.stats = group, stats
        ctx.meta = [(t.shape, t.dtype, t.device) for t in tensors]
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        # every rank must contribute a bucket of the same size: an output the local loss did not touch counts as zeros
        parts = [(g if g is not None else torch.zeros(sh, dtype=dt, device=dv)).reshape(-1).float()
                 for g, (sh, dt, dv) in zip(grads, ctx.meta)]
        flat = torch.cat(parts)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=ctx.group)
        if ctx.stats is not None:
            ctx.stats["all_reduce_calls"] = ctx.stats.get("all_reduce_calls", 0) + 1
            ctx.stats["all_reduce_bytes"] = flat.numel() * 4
        out, o = [], 0
        for (sh, dt, dv) in ctx.meta:
            n = 1
            for d in sh:
                n *= int(d)
            out.append(flat[o:o + n].reshape(sh).to(dt))
            o += n
        return (None, None) + tuple(out)


class ShardedRenderWrapper(torch.nn.Module):
    """Callable like the reference's DataParallel(_RenderWrapper, dim=1): every rank passes the
    same rays (SB,B,8); rank r renders rays[:, lo_r:hi_r]; every rank returns the full result
    (tuple (rgb, depth) for simple_output, else the nested dict).  All outputs of a call travel in ONE
    all_gather: they are packed along the last axis ((rgb | depth) = 16 B/ray for simple_output).

    Training (grad enabled, as train/train.py:75,199-215 does through DataParallel): the rank's own shard of the outputs
    keeps its autograd graph inside the gathered result (the other ranks' columns are constants), so the trainer's loss --
    computed identically on every rank from the full outputs -- back-propagates through the local rays only, and the partial
    gradients of both ResnetFCs and of `encoder.latent` are summed over the ranks by ONE all_reduce of one flattened bucket
    (`_BucketAllReduce`, entered by the renderer through `net._grad_sync`).  Every rank ends the step with the full
    gradient, exactly what a single process would have computed (tests/test_dist_gloo.py), and steps its own optimizer."""

    def __init__(self, wrapped, group=None):
        super().__init__()
        self.wrapped = wrapped
        self.group = group
        self.comm_stats = {}

    def _grad_sync(self, latent, params):
        ts = list(params) + ([latent] if (latent is not None and latent.requires_grad) else [])
        out = _BucketAllReduce.apply(self.group, self.comm_stats, *ts)
        n = len(params)
        return (out[n] if len(out) > n else latent), list(out[:n])

    def _empty_outputs(self, rays, rend, want_weights):
        """what the wrapped module returns for a (SB, 0, 8) ray batch, without launching anything (the kernels reject R = 0)"""
        SB, dev = rays.shape[0], rays.device

        def part(K):
            d = {"rgb": torch.zeros(SB, 0, 3, device=dev), "depth": torch.zeros(SB, 0, device=dev)}
            if want_weights and not getattr(self.wrapped, "simple_output", False):
                d["weights"] = torch.zeros(SB, 0, K, device=dev)
            return d
        Kc = rend.n_coarse
        res = {"coarse": part(Kc)}
        if rend.using_fine:
            res["fine"] = part(Kc + rend.n_fine)
        if getattr(self.wrapped, "simple_output", False):
            last = res["fine"] if rend.using_fine else res["coarse"]
            return last["rgb"], last["depth"]
        return res

    def forward(self, rays, want_weights=False):
        net = getattr(self.wrapped, "net", None)
        training = (net is not None and torch.is_grad_enabled()
                    and (any(p.requires_grad for p in net.parameters())
                         or (torch.is_tensor(getattr(getattr(net, "encoder", None), "latent", None)) and net.encoder.latent.requires_grad)))
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        B = rays.shape[1]
        if training and hasattr(net, "fused_supported") and not net.fused_supported():
            # the step's one gradient all-reduce is entered from the fused path's autograd Function; a network on the composed path
            # (a model conf / ResnetFC shape outside the shipped one) would train every rank on its own shard without it
            raise NotImplementedError("ShardedRenderWrapper: sharded TRAINING covers the fused network (the shipped model conf); a "
                                      "composed-path network renders sharded, but its gradients are not all-reduced here")
        if training and B < world:
            # a rank without rays would never enter the renderer's autograd Function, i.e. never join the step's one
            # all_reduce, and its peers would block in it: refuse up front, identically on every rank
            raise ValueError(f"ShardedRenderWrapper: training needs at least one ray per rank (rays per object B = {B}, "
                             f"world size = {world}); use a ray batch >= the number of ranks")
        bounds = [shard_bounds(B, r, world) for r in range(world)]
        sizes = [hi - lo for lo, hi in bounds]
        lo, hi = bounds[rank]
        rend = getattr(self.wrapped, "renderer", None)
        if rend is not None and hasattr(rend, "ray_id_offset"):
            # counter-based draws are keyed by the GLOBAL ray id: the sharded image equals the unsharded one (every rank
            # must run with the same torch seed; the per-renderer call counter advances in lock-step)
            rend.ray_id_offset, rend.ray_id_stride = lo, B
        if training:
            net._grad_sync = self._grad_sync  # render_autograd routes (latent, parameters) through the bucket all-reduce
        try:
            if hi == lo and rend is not None:   # fewer rays than ranks (inference): this rank contributes empty columns
                local = self._empty_outputs(rays, rend, want_weights)
            else:
                local = self.wrapped(rays[:, lo:hi].contiguous(), want_weights=want_weights)
        finally:
            if training:
                net._grad_sync = None
            if rend is not None and hasattr(rend, "ray_id_offset"):
                rend.ray_id_offset, rend.ray_id_stride = 0, 0
        # flatten the outputs to (SB, b, width) columns, pack, gather once, unpack
        if isinstance(local, tuple):
            leaves = [(None, i, t) for i, t in enumerate(local)]
        else:
            leaves = [(k, kk, vv) for k, v in local.items() for kk, vv in v.items()]
        def _width(t):  # explicit: reshape(..., -1) is ambiguous for an empty shard
            w = 1
            for d in t.shape[2:]:
                w *= int(d)
            return w
        cols = [t.reshape(t.shape[0], t.shape[1], _width(t)) for _, _, t in leaves]
        widths = [c.shape[-1] for c in cols]
        packed = torch.cat(cols, dim=-1)
        full = _gather_dim1(packed.detach(), sizes, self.group)
        if training and packed.requires_grad:  # this rank's columns keep their graph; the other ranks' are constants
            full = torch.cat([full[:, :lo], packed, full[:, hi:]], dim=1)
        outs, o = [], 0
        for (_, _, t), w in zip(leaves, widths):
            outs.append(full[..., o:o + w].reshape((t.shape[0], B) + tuple(t.shape[2:])))
            o += w
        if isinstance(local, tuple):
            return tuple(outs)
        res = {}
        for (k, kk, _), v in zip(leaves, outs):
            res.setdefault(k, {})[kk] = v
        return res
