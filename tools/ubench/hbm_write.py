#!/usr/bin/env python3
"""What a write-dominated stream reaches on this GPU: fill (write only), copy (read + write) and a read-only reduction over the
size of the DTU feature grid (3 x 150 x 200 x 512 fp32 = 184 MB), timed with HIP events.  Yardstick for pnr_pyramid_to_latent."""
import torch

dev = torch.device("cuda:0")
n = 3 * 150 * 200 * 512
a = torch.empty(n, device=dev)
b = torch.randn(n, device=dev)


def t(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for name, fn, by in (("fill_ (write only)", lambda: a.fill_(1.0), n * 4), ("zero_ (memset)", lambda: a.zero_(), n * 4),
                     ("copy_ (read + write)", lambda: a.copy_(b), 2 * n * 4), ("sum (read only)", lambda: b.sum(), n * 4)):
    dt = t(fn)
    print(f"{name:22s}: {dt * 1e6:7.1f} us  {by / 1e6:6.1f} MB  {by / dt / 1e12:5.2f} TB/s")
