import sys, time, torch
sys.path.insert(0, '.')
from pixelnerf_amd import ops
dev = torch.device('cuda:0')
rows = 49152
for name, dt, prec, mag in (("f16 normal", torch.float16, 0, 1.0), ("f16 tiny", torch.float16, 0, 1e-6), ("f16 1e-3", torch.float16, 0, 1e-3), ("bf16 normal", torch.bfloat16, 1, 1.0), ("bf16 tiny", torch.bfloat16, 1, 1e-6)):
    dY = (torch.randn(rows, 512, device=dev) * mag).to(dt)
    X = torch.randn(rows, 512, device=dev).to(dt).relu()
    for _ in range(2): ops.weight_grad(dY, X, prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.weight_grad(dY, X, prec)
    torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 10
    ref = dY.float().t() @ X.float()
    got, db = ops.weight_grad(dY, X, prec)
    print(f"{name:12s}: {dt_*1e3:7.3f} ms   rel err {float((got-ref).norm()/ref.norm()):.2e}  db err {float((db - dY.float().sum(0)).norm()/dY.float().sum(0).norm()):.2e}")
