#!/usr/bin/env python3
"""pnr_weight_grad_batched at PNR_PREC_F16X3 on the training step's shape: 14 jobs (ten fc + three lin_z of 512 columns, lin_in of 64),
rows = 32768 (coarse pass of config 5) and 49152 (fine pass); HIP events, us per call (dw kernel + reduction), max error of one job
against fp64, a checksum of all outputs (equal across forms when the partial sums are formed in the same order).
A/B: PNR_DW_FORM=8wave | (unset: the one-wave-per-SIMD kernel); variant libraries through PIXELNERF_HIP_LIB (profiles/r06_dw_split_notes.md)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pixelnerf_amd import ops, _lib
dev = torch.device("cuda:0")
tag = os.environ.get("PNR_DW_FORM", "wide") + ":" + os.path.basename(os.environ.get("PIXELNERF_HIP_LIB", "product"))
for rows in [int(a) for a in sys.argv[1:]] or [32768, 49152]:
    g = torch.Generator(device=dev).manual_seed(rows)
    def pair(cols, scale):
        v = torch.randn(rows, cols, device=dev, generator=g) * scale
        h = v.half()
        return torch.stack([h, (v - h.float()).half()])
    jobs = []
    for j in range(13):
        jobs.append((pair(512, 1e-2), pair(512, 1.0), True, j < 10))
    jobs.append((pair(512, 1e-2), pair(64, 1.0), True, False, 64, 42))
    outs = ops.weight_grad_batched(jobs, _lib.PREC_F16X3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        outs = ops.weight_grad_batched(jobs, _lib.PREC_F16X3)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    # job 10: natural column order, storage-order rows (the permutation is the kernel's business: compare as sets per row is not
    # possible -- use job 13's db and job 10's plain Frobenius norm instead, plus an exact reference of job 10 through the library's own
    # feature order on a permutation-invariant quantity)
    dY, X = jobs[10][0].double().sum(0), jobs[10][1].double().sum(0)
    ref = dY.t() @ X
    dW, db = outs[10]
    err_norm = abs(float(dW.double().norm()) - float(ref.norm())) / float(ref.norm())
    err_db = abs(float(db.double().sum()) - float(dY.sum())) / max(abs(float(dY.sum())), 1e-30)
    chk = sum(float(w.double().sum()) + float(b.double().sum()) for w, b in outs)
    flop = 3 * 2 * rows * 512 * (13 * 512 + 64)
    print(f"dw[{tag}] rows {rows}: {us:8.1f} us per call ({flop / us / 1e6:6.1f} TFLOP/s of executed MFMAs)  |dW| rel err {err_norm:.2e}  sum(db) rel err {err_db:.2e}  checksum {chk:.12e}", flush=True)
