#!/usr/bin/env python3
"""
BASELINE config (4): one full-resolution DTU view (400x300 = 120 000 rays, 3 source views,
64+128 samples) rendered by N ranks (strong scaling):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        tools/render_sharded.py [--scene dtu] [--backend nccl]

Rank 0 "encodes" (synthetic feature grid), broadcast_encoded() ships the 176 MiB grid once, every
rank renders its contiguous slice of the rays through bind_parallel(net, gpus) ->
ShardedRenderWrapper, which all-gathers (rgb, depth).  Prints per-call time = broadcast + render +
gather (max over ranks) and rays/s.  With one process it degenerates to the single-GPU render.
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from testdata import synthetic  # noqa: E402
from pixelnerf_amd.dist import broadcast_encoded  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="dtu")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--prec", default="f16")
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    scene, meta, net, renderer, _ = bench.build(dev, args.prec, args.scene)
    rays = synthetic.target_rays(meta).to(dev)  # (1, H*W, 8), identical on every rank
    if world > 1 and rank != 0:  # only rank 0 holds the encoded scene before the broadcast
        net.encoder.latent = torch.zeros(1, 1, 1, 1, device=dev)
    render_par = renderer.bind_parallel(net, list(range(world)) if world > 1 else None, simple_output=True).eval()

    def call():
        if world > 1:
            broadcast_encoded(net, src=0)
        with torch.no_grad():
            return render_par(rays)

    for _ in range(2):
        rgb, depth = call()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.iters):
        rgb, depth = call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.iters
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        R = rays.shape[1]
        assert rgb.shape == (1, R, 3) and torch.isfinite(rgb).all()
        print(f"{args.scene}: {meta['W']}x{meta['H']} = {R} rays, NS={scene['NS']}, {world} rank(s): "
              f"{dt*1e3:.1f} ms per image (broadcast+render+gather)  {R/dt/1e3:.1f} k rays/s", flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
