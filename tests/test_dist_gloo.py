"""world_size-2 gloo tests (CPU) of the multi-GPU plumbing: ray sharding on dim 1, uneven
all_gather, and the one-shot broadcast of the encoded scene.  The renderer itself is replaced
by a deterministic pure function of the rays so the test exercises only the distributed logic
(the HIP kernels are covered by the -m gpu tests)."""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pixelnerf_amd.dist import ShardedRenderWrapper, broadcast_encoded, shard_bounds


def test_shard_bounds_partition():
    for n in (0, 1, 5, 8, 4096, 120000):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


class FakeWrapped(torch.nn.Module):
    def __init__(self, simple):
        super().__init__()
        self.simple = simple

    def forward(self, rays, want_weights=False):
        rgb = rays[..., :3] * 2 + rays[..., 6:7]
        depth = rays[..., 3:6].sum(-1)
        if self.simple:
            return rgb, depth
        w = rays[..., :5].cumsum(-1)
        return {"coarse": {"rgb": rgb, "depth": depth, "weights": w}, "fine": {"rgb": rgb + 1, "depth": depth * 2}}


def _worker(rank, world, port, B):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        rays = torch.randn(2, B, 8)
        for simple in (True, False):
            full = FakeWrapped(simple)(rays)
            got = ShardedRenderWrapper(FakeWrapped(simple))(rays, want_weights=not simple)
            if simple:
                assert all(torch.equal(a, b) for a, b in zip(full, got))
            else:
                for k in full:
                    for kk in full[k]:
                        assert torch.equal(full[k][kk], got[k][kk]), (k, kk)
        # encoded-scene broadcast: rank 0 holds the state, rank 1 starts empty
        enc = SimpleNamespace(latent=torch.zeros(1, 1, 1, 1), latent_scaling=torch.zeros(2))
        net = SimpleNamespace(encoder=enc, poses=torch.zeros(1, 3, 4), focal=torch.zeros(1, 2), c=torch.zeros(1, 2),
                              image_shape=torch.zeros(2), num_views_per_obj=1, num_objs=0)
        g = torch.Generator().manual_seed(3)
        ref = dict(latent=torch.randn(4, 6, 5, 7, generator=g), poses=torch.randn(4, 3, 4, generator=g),
                   focal=torch.randn(2, 2, generator=g), c=torch.randn(1, 2, generator=g),
                   image_shape=torch.tensor([64.0, 48.0]), ls=torch.tensor([2.1, 2.2]))
        if rank == 0:
            enc.latent, enc.latent_scaling = ref["latent"].clone(), ref["ls"].clone()
            net.poses, net.focal, net.c = ref["poses"].clone(), ref["focal"].clone(), ref["c"].clone()
            net.image_shape, net.num_views_per_obj, net.num_objs = ref["image_shape"].clone(), 2, 2
        calls = {"broadcast": 0, "all_gather": 0}
        real_b, real_g = dist.broadcast, dist.all_gather
        dist.broadcast = lambda *a, **k: (calls.__setitem__("broadcast", calls["broadcast"] + 1), real_b(*a, **k))[1]
        dist.all_gather = lambda *a, **k: (calls.__setitem__("all_gather", calls["all_gather"] + 1), real_g(*a, **k))[1]
        try:
            # north_star: ONE broadcast of the encoded grid (receivers know the dataset's grid shape) ...
            broadcast_encoded(net, src=0, latent_shape=(4, 6, 5, 7))
            assert calls["broadcast"] == 1, calls
            # ... and ONE gather of the packed outputs per render call, whatever the output structure
            ShardedRenderWrapper(FakeWrapped(False))(rays, want_weights=True)
            assert calls["all_gather"] == 1, calls
        finally:
            dist.broadcast, dist.all_gather = real_b, real_g
        assert torch.equal(enc.latent, ref["latent"]) and torch.equal(net.poses, ref["poses"])
        assert (net.num_views_per_obj, net.num_objs) == (2, 2)
        if rank != 0:  # shape-discovery form (header first): wipe and receive again
            enc.latent, net.poses, net.num_views_per_obj = torch.zeros(1, 1, 1, 1), torch.zeros(1, 3, 4), 1
        broadcast_encoded(net, src=0)
        assert torch.equal(enc.latent, ref["latent"]) and torch.equal(net.poses, ref["poses"])
        assert torch.equal(net.focal, ref["focal"]) and torch.equal(net.c, ref["c"])
        assert torch.equal(net.image_shape, ref["image_shape"]) and torch.equal(enc.latent_scaling, ref["ls"])
        assert (net.num_views_per_obj, net.num_objs) == (2, 2)
        # the flat 1 -> (N-1) fan-out form of the same transfer (bench.py --bcast flat): point-to-point only,
        # NO collective, same state on the receivers
        if rank != 0:
            enc.latent, net.poses, net.num_views_per_obj = torch.zeros(1, 1, 1, 1), torch.zeros(1, 3, 4), 1
        calls["broadcast"] = 0
        dist.broadcast = lambda *a, **k: (calls.__setitem__("broadcast", calls["broadcast"] + 1), real_b(*a, **k))[1]
        try:
            broadcast_encoded(net, src=0, latent_shape=(4, 6, 5, 7), algo="flat")
        finally:
            dist.broadcast = real_b
        assert calls["broadcast"] == 0
        assert torch.equal(enc.latent, ref["latent"]) and torch.equal(net.poses, ref["poses"])
        assert torch.equal(net.focal, ref["focal"]) and (net.num_views_per_obj, net.num_objs) == (2, 2)
        # channel-last transfer (layout="nhwc"): the receivers get the grid in the layout the fused kernels read, installed as
        # the encoder's cached channel-last copy, and `encoder.latent` is its (N,C,H,W)-shaped view -- same values, no transpose
        if rank != 0:
            enc.latent, net.poses, net.num_views_per_obj = torch.zeros(1, 1, 1, 1), torch.zeros(1, 3, 4), 1
        enc.latent_nhwc = lambda: enc.latent.permute(0, 2, 3, 1).contiguous()
        broadcast_encoded(net, src=0, latent_shape=(4, 6, 5, 7), layout="nhwc")
        assert tuple(enc.latent.shape) == (4, 6, 5, 7) and torch.equal(enc.latent, ref["latent"]) and torch.equal(net.poses, ref["poses"])
        if rank != 0:
            key, nhwc = enc._nhwc
            assert nhwc.is_contiguous() and torch.equal(nhwc, ref["latent"].permute(0, 2, 3, 1))
            assert key == (enc.latent.data_ptr(), enc.latent._version, (4, 6, 5, 7)) and nhwc.data_ptr() == enc.latent.data_ptr()
        # strong-scaling placement (bench.py --workload dtu): contiguous shards of ONE image, gathered with padding, give
        # back the image in ray order
        Rimg = 11
        img = torch.arange(Rimg * 4, dtype=torch.float32).reshape(Rimg, 4)
        lo, hi = shard_bounds(Rimg, rank, world)
        sizes = [shard_bounds(Rimg, r, world)[1] - shard_bounds(Rimg, r, world)[0] for r in range(world)]
        part = img[lo:hi]
        if part.shape[0] < max(sizes):
            part = torch.cat([part, part.new_zeros(max(sizes) - part.shape[0], 4)])
        bufs = [torch.empty_like(part) for _ in range(world)] if rank == 0 else None
        dist.gather(part.contiguous(), bufs, dst=0)
        if rank == 0:
            assert torch.equal(torch.cat([b[:n] for b, n in zip(bufs, sizes)]), img)
    finally:
        dist.destroy_process_group()


class FakeTrainNet(torch.nn.Module):
    """stands in for PixelNeRFNet on the CPU: two "networks" of parameters + an encoder.latent, and the `_grad_sync` protocol
    the renderer follows (autograd.render_autograd): parameters and latent pass through the hook before they are used"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(5)
        self.a = torch.nn.Parameter(torch.randn(8, 3, generator=g))
        self.b = torch.nn.Parameter(torch.randn(3, generator=g))
        self.c = torch.nn.Parameter(torch.randn(8, generator=g))   # touched by the depth output only
        self.encoder = SimpleNamespace(latent=torch.randn(4, 5, generator=g).requires_grad_(True))
        self._grad_sync = None


class FakeTrainWrapped(torch.nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net

    def forward(self, rays, want_weights=False):
        net = self.net
        latent, params = net.encoder.latent, [net.a, net.b, net.c]
        if net._grad_sync is not None:
            latent, params = net._grad_sync(latent, params)
        a, b, c = params
        rgb = torch.tanh(rays @ a + b) * latent.sum()
        depth = (rays * c).sum(-1) + latent[0, 0] * rays[..., 0]
        return {"coarse": {"rgb": rgb, "depth": depth}}


def _train_worker(rank, world, port, B):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(1)
        rays = torch.randn(2, B, 8)
        gt = torch.randn(2, B, 3)

        def loss_of(out):  # train/train.py:199-215 shape: a mean over ALL rays, computed from the full outputs
            return ((out["coarse"]["rgb"] - gt) ** 2).mean() + 0.1 * (out["coarse"]["depth"] ** 2).mean()

        ref_net = FakeTrainNet()
        loss_of(FakeTrainWrapped(ref_net)(rays)).backward()  # single process: the whole ray batch
        net = FakeTrainNet()
        par = ShardedRenderWrapper(FakeTrainWrapped(net))
        out = par(rays)
        assert out["coarse"]["rgb"].shape == (2, B, 3)
        loss = loss_of(out)
        loss.backward()
        # ONE all_reduce of one bucket holding every parameter gradient + the latent gradient ...
        assert par.comm_stats["all_reduce_calls"] == 1
        assert par.comm_stats["all_reduce_bytes"] == 4 * (8 * 3 + 3 + 8 + 4 * 5)
        # ... after which EVERY rank holds the single-process gradient
        for got, ref in ((net.a.grad, ref_net.a.grad), (net.b.grad, ref_net.b.grad), (net.c.grad, ref_net.c.grad),
                         (net.encoder.latent.grad, ref_net.encoder.latent.grad)):
            assert torch.allclose(got, ref, rtol=1e-5, atol=1e-7), (got - ref).abs().max()
        assert net._grad_sync is None  # the hook is only installed for the duration of the call
        with torch.no_grad():  # inference through the same wrapper: no bucket, no graph
            o2 = par(rays)
        assert not o2["coarse"]["rgb"].requires_grad and par.comm_stats["all_reduce_calls"] == 1
        # fewer rays than ranks: a rank without rays would never join the step's all_reduce -> refused up front on every rank
        with pytest.raises(ValueError, match="at least one ray per rank"):
            par(rays[:, :1])
        with torch.no_grad():  # ... while inference just gathers an empty shard
            o3 = par(rays[:, :1])
        assert o3["coarse"]["rgb"].shape == (2, 1, 3)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 7), (2, 64), (4, 7), (4, 13)])  # 4 ranks: shards of 2,2,2,1 and 4,3,3,3 rays
def test_sharded_training_gradients_equal_single_process(world, B):
    """the differentiable multi-process path (reference: DataParallel training, train/train.py:75; src/render/nerf.py:367-371):
    sharded forward, loss on the gathered outputs, ONE bucketed gradient all-reduce -> single-process gradients on every rank"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_train_worker, args=(world, port, B), nprocs=world, join=True)


@pytest.mark.parametrize("world,B", [(2, 7), (2, 64), (4, 7), (4, 64)])  # uneven and even splits
def test_sharded_render_and_scene_broadcast(world, B):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, B), nprocs=world, join=True)


def test_eight_way_shards_of_the_dtu_image_and_the_weak_batch():
    """the placements the 8-GPU scaling run uses (bench.py): 120 000 DTU rays -> 15 000 per rank; an uneven image; a batch with
    fewer rays than ranks -- contiguous, ordered, covering, sizes within one ray of each other (DataParallel's dim-1 split)"""
    for n, world in ((120000, 8), (120001, 8), (16384, 8), (5, 8), (65536 * 8, 8), (120000, 4)):
        b = [shard_bounds(n, r, world) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes
    assert shard_bounds(120000, 3, 8) == (45000, 60000)
