"""
Multi-GPU rendering: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm).

The reference's only parallelism is single-process torch.nn.DataParallel(dim=1)
(src/render/nerf.py:367-371), which on EVERY call re-broadcasts the whole wrapper (ResNet-34,
both MLPs and the encoded feature grid), scatters the rays and gathers the outputs.  Rays are
independent units, so here:

  * weights live on every rank (loaded once);
  * after `net.encode()` on one rank, `broadcast_encoded()` ships the feature grid with ONE
    collective (+ one tiny metadata message): sn64 2 MiB, srn 16 MiB, DTU 176 MiB;
  * each rank renders a contiguous slice of the rays on dim 1 (`shard_bounds`), exactly the
    split DataParallel(dim=1) makes;
  * results come back with one all_gather of (rgb, depth) = 16 B/ray (`ShardedRenderWrapper`).

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


def broadcast_encoded(net, src=0, group=None):
    """Make rank `src`'s encode() state current on every rank: one broadcast of the feature grid
    and one of the packed camera metadata (poses, focal, c, image_shape)."""
    dev = net.poses.device
    if dist.get_rank(group) == src:
        lat = net.encoder.latent
        hdr = torch.tensor([lat.shape[0], lat.shape[1], lat.shape[2], lat.shape[3], net.num_views_per_obj,
                            net.num_objs, net.focal.shape[0], net.c.shape[0]], dtype=torch.int64, device=dev)
    else:
        hdr = torch.zeros(8, dtype=torch.int64, device=dev)
    dist.broadcast(hdr, src, group=group)
    NV, C, Hl, Wl, NS, SB, nf, nc = [int(v) for v in hdr.tolist()]
    n_meta = NV * 12 + nf * 2 + nc * 2 + 2 + 2
    if dist.get_rank(group) == src:
        meta = torch.cat([net.poses.reshape(-1).float(), net.focal.reshape(-1).float(), net.c.reshape(-1).float(),
                          net.image_shape.reshape(-1).float(), net.encoder.latent_scaling.reshape(-1).float()])
        lat = net.encoder.latent.contiguous()
    else:
        meta = torch.empty(n_meta, dtype=torch.float32, device=dev)
        lat = torch.empty((NV, C, Hl, Wl), dtype=torch.float32, device=dev)
    dist.broadcast(lat, src, group=group)      # THE feature-grid broadcast
    dist.broadcast(meta, src, group=group)
    if dist.get_rank(group) != src:
        o = 0
        net.encoder.latent = lat
        net.poses = meta[o:o + NV * 12].reshape(NV, 3, 4).clone(); o += NV * 12
        net.focal = meta[o:o + nf * 2].reshape(nf, 2).clone(); o += nf * 2
        net.c = meta[o:o + nc * 2].reshape(nc, 2).clone(); o += nc * 2
        net.image_shape = meta[o:o + 2].clone(); o += 2
        net.encoder.latent_scaling = meta[o:o + 2].clone()
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


class ShardedRenderWrapper(torch.nn.Module):
    """Callable like the reference's DataParallel(_RenderWrapper, dim=1): every rank passes the
    same rays (SB,B,8); rank r renders rays[:, lo_r:hi_r]; every rank returns the full result
    (tuple (rgb, depth) for simple_output, else the nested dict)."""

    def __init__(self, wrapped, group=None):
        super().__init__()
        self.wrapped = wrapped
        self.group = group

    def forward(self, rays, want_weights=False):
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        B = rays.shape[1]
        bounds = [shard_bounds(B, r, world) for r in range(world)]
        sizes = [hi - lo for lo, hi in bounds]
        lo, hi = bounds[rank]
        local = self.wrapped(rays[:, lo:hi].contiguous(), want_weights=want_weights)

        def gather(t):
            return _gather_dim1(t, sizes, self.group)

        if isinstance(local, tuple):
            return tuple(gather(t) for t in local)
        return {k: {kk: gather(vv) for kk, vv in v.items()} for k, v in local.items()}
