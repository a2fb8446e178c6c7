#!/usr/bin/env python3
"""
bench.py -- rays/sec of the pixelNeRF volume-rendering hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--prec f16x3|f16] [--rays R]
                    [--workload sn64|dtu] [--bcast tree|flat]

Headline workload (BASELINE.json configs[1], the configuration the metric is quoted on): sn64 NMR geometry, 64x64 target
views, 1 source view, 64 coarse + 128 fine samples (112 importance + 16 depth), separate coarse / fine ResnetFC
(d_hidden 512, 5 blocks), synthetic feature grid and random-init weights (no datasets / checkpoints offline).  One step =
one `render_par(rays)` call, exactly what eval/eval.py:277 times in the reference: R = 65536 rays (16 target views) per GPU,
through NeRFRenderer/_RenderWrapper (noise draws included, lin_z re-folded into the grid inside every step).
`value` = rays rendered by all ranks / wall time of the K timed steps (inputs resident in HBM).

PRECISION.  The reference computes in fp32 (nn.Linear, no autocast: src/model/resnetfc.py:147,175-180,55-62).  The timed
headline therefore runs `precision="f16x3"` -- the fp32-CLASS fused kernel (pnr_split.hip: every operand a (head, tail)
pair of fp16, three f16 MFMAs per product, fp32 accumulation and tables; per-point |rgb| <= 2e-5 against the reference's
goldens, tests/test_hip_split.py) -- and `dtype` says so.  The 16-bit-operand kernel (`precision="f16"`, PSNR >= 52 dB
against the reference) is timed in a second region of the same length and reported as a PEER block `f16_path`
with its own roofline: faster, but narrower arithmetic than the reference, so it is not `value`.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): weak scaling -- every rank renders its own R rays; each
timed step also contains the single feature-grid broadcast from rank 0 and the final gather of (rgb, depth) to rank 0
(SURVEY.md 8e).  The same run then times BASELINE configs[3] in its strong-scaling form (`extra.strong_dtu`: ONE DTU
400x300 image, 3 source views, 176 MiB grid, its 120 000 rays sharded contiguously over the N ranks; step = grid broadcast
+ render + gather).  `--workload dtu` makes that form the headline instead (`scaling: "strong"`).

The JSON line also carries
  roofline     : the fused network kernel (>= 99 % of the work) against the dense f16 MFMA peak: algorithmic FLOP of the
                 launches in the timed region / their HIP-event time on the launch stream; executed_mfma_tflops counts the
                 MFMAs really issued (3 per product for f16x3, lin_z folded away); traffic = bytes per launch at the L2 <-> fabric
                 interface, MEASURED IN THIS RUN by two rocprofv3 --pmc passes around a 2-step child of the same command (the
                 committed PMC profile, stamped with the kernel-source hash, is the fallback and rides along for comparison);
  torch_eager_gpu_baseline : the same algorithm in eager PyTorch-ROCm fp32 on this GPU (oracle restatement), in the reference's
                 execution shape -- 50 000-ray batches, 50 000-point model calls (eval/eval.py:137,264; nerf.py:190-216) -- and,
                 under ..._unchunked_16384, as one model call per pass; speedup_vs_torch_eager_gpu refers to the former;
  cpu_baseline : the CPU restatement of the reference (oracle, kind "port", F.grid_sample like the reference) timed on this
                 host's cores on a bounded sample of the same workload, rank 0 / N=1 only; cpu_baseline_config1 =
                 BASELINE configs[0] verbatim (32 coarse, no fine pass, one 4096-ray call, CPU);
  psnr_db      : PSNR of the timed path's render vs that CPU render on the sample (identical rays, weights, grid, noise);
  latency_4096_rays_ms, encode_ms : SURVEY 8d's single-image latency and the encoder time (ResNet-34 trunk in
                 PyTorch-ROCm + pnr_pyramid_to_latent), both outside `value`.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_POINT_VIEW = 4.7616e6   # lin_in + 3 lin_z + 3 blocks, per (point, view)   SURVEY.md 8a
FLOP_PER_POINT_POOLED = 2.1012e6  # 2 blocks + lin_out, per point
FLOP_LIN_Z_PER_POINT_VIEW = 3 * 2 * 512 * 512  # the three lin_z layers (folded into per-texel tables at inference)
PEAK_TFLOPS = 2500.0  # dense f16 / bf16 MFMA peak, MI355X_MICROARCH.md (both kernels issue v_mfma_f32_32x32x16_f16)
MFMAS_PER_PRODUCT = {"f16": 1, "bf16": 1, "f16x3": 3}
KERNEL_OF = {"f16": "pnr::eval_kernel", "bf16": "pnr::eval_kernel", "f16x3": "pnr::eval_split_kernel"}
KERNEL_SOURCES = {"f16": ["pnr_mlp.hip", "pnr_device.h", "pnr_layout.h"], "bf16": ["pnr_mlp.hip", "pnr_device.h", "pnr_layout.h"],
                  "f16x3": ["pnr_split.hip", "pnr_device.h", "pnr_layout.h"]}
PMC_PROFILE = {"f16": os.path.join("profiles", "r06_bench_f16_pmc_eval_kernel.json"),
               "f16x3": os.path.join("profiles", "r06_bench_f16x3_pmc_eval_split_kernel.json")}
DTYPE_NOTE = {"f16x3": "fp32-class: (head, tail) fp16 operand pairs, 3 f16 MFMAs per product, fp32 accumulate, fp32 tables; "
                       "per-point |rgb| <= 2e-5 vs the reference (the reference's own arithmetic class)",
              "f16": "fp16 MFMA operands, fp32 accumulate: narrower than the reference's fp32 (PSNR >= 52 dB bar)",
              "bf16": "bf16 MFMA operands (experiment flag)"}


def build(dev, prec, scene_name="sn64"):
    from testdata import synthetic
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util.conf import default_model_conf

    scene, meta = synthetic.make_scene(scene_name)
    net = make_model(default_model_conf(), precision=prec).to(dev).eval()
    mc, mf = synthetic.make_mlp_params(11), synthetic.make_mlp_params(12)
    net.mlp_coarse.load_state_dict(mc)
    net.mlp_fine.load_state_dict(mf)
    lat = scene["latent"].to(dev)
    net.encoder.latent = lat
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32, device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = scene["SB"], scene["NS"]
    renderer = NeRFRenderer(n_coarse=64, n_fine=128, n_fine_depth=16, depth_std=0.01,
                            white_bkgd=meta["white_bkgd"], lindisp=False).to(dev).eval()
    return scene, meta, net, renderer, (mc, mf)


def make_rays(meta, R, rank):
    """R rays: whole 64x64 target views on the sn64 camera circle (different views per rank)."""
    from testdata import synthetic
    n_img = (R + meta["W"] * meta["H"] - 1) // (meta["W"] * meta["H"])
    poses = torch.stack([synthetic.pose_spherical(75.0 + 360.0 * (i + rank * n_img) / (n_img * 8 + 1), -20.0,
                                                  meta["radius"]) for i in range(n_img)])
    rays = synthetic.gen_rays(poses, meta["W"], meta["H"], meta["focal"], meta["z_near"], meta["z_far"], c=meta["c"])
    return rays.reshape(-1, 8)[:R].contiguous()


def flop_per_ray(NS, n_coarse=64, n_fine=128):
    evals = n_coarse + ((n_coarse + n_fine) if n_fine > 0 else 0)
    return evals * (FLOP_PER_POINT_VIEW * NS + FLOP_PER_POINT_POOLED)


def executed_fraction(NS, fold):
    """share of the algorithmic FLOP that is issued as per-sample GEMMs (lin_z leaves the stream when folded)"""
    per_pt = FLOP_PER_POINT_VIEW * NS + FLOP_PER_POINT_POOLED
    return 1.0 - (NS * FLOP_LIN_Z_PER_POINT_VIEW) / per_pt if fold else 1.0


def cpu_baseline(scene, mlps, rays_sample, noise, threads=None, n_coarse=64, n_fine=128, n_fine_depth=16):
    """Oracle (CPU restatement of the reference, F.grid_sample like the reference) on a bounded sample; rays/s + render."""
    from oracle import pnr_oracle as O
    if threads:
        torch.set_num_threads(threads)
    prev, O.USE_GRID_SAMPLE = O.USE_GRID_SAMPLE, True  # the reference's op (src/model/encoder.py:96-109)
    try:
        with torch.no_grad():
            t0 = time.perf_counter()
            out = O.render(scene, mlps[0], mlps[1], rays_sample[None], noise, n_coarse, n_fine, n_fine_depth, white_bkgd=True)
            dt = time.perf_counter() - t0
    finally:
        O.USE_GRID_SAMPLE = prev
    return rays_sample.shape[0] / dt, dt, out


def kernel_source_sha16(prec="f16"):
    """Identity of a fused network kernel: hash of the sources it is compiled from."""
    import hashlib
    h = hashlib.sha256()
    for f in KERNEL_SOURCES[prec]:
        h.update(open(os.path.join(ROOT, "pixel-nerf_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(prec):
    """HBM-side bytes per fused-kernel launch from the committed rocprofv3 PMC passes of THIS command
    (tools/collect_pmc.sh: separate --pmc runs for FETCH_SIZE and WRITE_SIZE, KiB units, FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 correction; average over the coarse and fine launches, like `achieved`).
    The profile carries the hash of the kernel sources it was measured on; a profile of a different kernel is
    refused (-> (None, reason)) instead of silently going stale."""
    rel = PMC_PROFILE.get(prec)
    if rel is None:
        return None, "no PMC profile for precision %s" % prec, None
    try:
        d = json.load(open(os.path.join(ROOT, rel)))
    except Exception:
        return None, "no PMC profile committed for this kernel (%s absent)" % rel, None
    if d.get("_kernel_source_sha16") != kernel_source_sha16(prec):
        return None, "%s was measured on kernel sources %s, this build is %s: refused as stale" % (
            rel, d.get("_kernel_source_sha16"), kernel_source_sha16(prec)), None
    return (2.0 * d["FETCH_SIZE"]["avg_per_launch"] + d["WRITE_SIZE"]["avg_per_launch"]) * 1024.0, None, d.get("_derived")


def measure_traffic_live(prec, timeout_s=150):
    """`roofline.traffic` measured IN THIS RUN: rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; GRBM_GUI_ACTIVE + the MFMA-busy pair --
    separate passes, no trace domain besides --kernel-trace, as MI355X_MICROARCH.md prescribes) around a 2-step child of this very
    command; per launch of the fused kernel, averaged over its coarse and fine launches like `achieved`; KiB units, FETCH_SIZE
    doubled (gfx950).  The third pass also yields the average shader clock DURING the kernel (GRBM_GUI_ACTIVE summed over the 8 XCDs
    / 8 / the launch's duration in that pass's kernel trace) and the MFMA-busy share: the pool's boxes sustain different clocks
    under this kernel's load, which is most of their 3-5 % spread in `value`.
    -> (bytes per launch, None, extras) or (None, reason, {}).  Never raises: the committed profile stays the fallback."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    tool = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(tool):
        return None, "rocprofv3 not found", {}
    kname = KERNEL_OF[prec].split("::")[-1]
    is_kernel = lambda n: kname + "<" in n or kname + "I" in n  # noqa: E731
    env = dict(os.environ, PIXELNERF_SATURATION_GUARD="off", TMPDIR="/tmp")  # every launch = the plain instantiation
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    got, extras = {}, {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as out:
            for name, counters, required in (("fetch", ["FETCH_SIZE"], True), ("write", ["WRITE_SIZE"], True),
                                             ("clock", ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"], False)):
                cmd = [tool, "--kernel-trace", "--pmc"] + counters + ["--output-format", "csv", "-d", out, "-o", "pmc_" + name, "--",
                       sys.executable, os.path.abspath(__file__), "--prec", prec, "--steps", "2", "--warmup", "1", "--no-peer", "--no-latency",
                       "--no-cpu-baseline", "--no-eager-baseline", "--no-f32-check", "--no-extras", "--no-live-pmc"]
                p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    p.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, 9)
                    if required:
                        return None, "rocprofv3 --pmc %s pass timed out after %d s" % (name, timeout_s), {}
                    continue
                per = {}
                for path in glob.glob(os.path.join(out, "**", "pmc_%s*counter_collection.csv" % name), recursive=True):
                    for row in csv.DictReader(open(path)):
                        if row["Counter_Name"] in counters and is_kernel(row["Kernel_Name"]):
                            d = per.setdefault(row["Counter_Name"], {})
                            d[row["Dispatch_Id"]] = d.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
                if required and counters[0] not in per:
                    return None, "the rocprofv3 --pmc %s pass (rc %s) recorded no %s launch" % (name, p.returncode, kname), {}
                for c, d in per.items():
                    got[c] = sum(d.values()) / len(d)
                if name == "clock" and "GRBM_GUI_ACTIVE" in got:
                    dur = []
                    for path in glob.glob(os.path.join(out, "**", "pmc_clock*kernel_trace.csv"), recursive=True):
                        for row in csv.DictReader(open(path)):
                            if is_kernel(row["Kernel_Name"]):
                                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)
                    if dur:
                        extras["shader_clock_ghz_during_kernel"] = got["GRBM_GUI_ACTIVE"] / 8.0 / (sum(dur) / len(dur)) / 1e9
                    if got.get("SQ_BUSY_CU_CYCLES"):
                        extras["mfma_busy_frac_measured_in_run"] = got["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * got["SQ_BUSY_CU_CYCLES"])
    except Exception as e:  # a profiler that is absent / refuses counters must not take the bench line down
        return None, "%s: %s" % (type(e).__name__, str(e)[:200]), {}
    return (2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024.0, None, extras


def roofline_block(prec, rays_rank0, steps, kern_ms, n_launch, NS, fold, elapsed, default_shape):
    """roofline of the dominant kernel: ALGORITHMIC flop of rank 0's launches in the timed region / their HIP-event time"""
    fpr = flop_per_ray(NS)
    ach = (rays_rank0 * steps * fpr) / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    traffic, why, derived = pmc_traffic_per_launch(prec) if default_shape else (None, "the committed profile is for 65536 sn64 rays, folded", None)
    ex = ach * executed_fraction(NS, fold) * MFMAS_PER_PRODUCT[prec]
    blk = {"bound": "mfma", "achieved": ach, "peak": PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS,
           "traffic": traffic,
           "traffic_source": None if traffic is None else "committed profile (%s), not measured in this run" % PMC_PROFILE.get(prec),
           "traffic_note": "bytes per launch at the L2<->fabric interface (Infinity-Cache hits included): 2*FETCH_SIZE + WRITE_SIZE "
                           "from %s (tools/collect_pmc.sh, stamped with the kernel-source hash %s); algorithmic HBM bytes per launch "
                           "are ~0.18 GB (z in, rgb-sigma out, weights + tables once); the excess is the weight stream (> 4 MB L2 per "
                           "XCD) re-fetched from the Infinity Cache once per tile pass and XCD%s" % (
                               PMC_PROFILE.get(prec), kernel_source_sha16(prec), "" if why is None else "; NULL because " + why),
           "kernel": KERNEL_OF[prec] + " (fused per-point network)", "launches": n_launch,
           "avg_launch_ms": kern_ms / max(n_launch, 1), "flop_per_ray": fpr,
           "kernel_time_frac_of_step": kern_ms * 1e-3 / elapsed,
           "executed_mfma_tflops": ex, "executed_mfma_frac_of_peak": ex / PEAK_TFLOPS,
           "mfmas_per_product": MFMAS_PER_PRODUCT[prec],
           "note": "achieved = ALGORITHMIC FLOP (SURVEY 8d: 1.757 GFLOP/ray at NS=1) / kernel time, against the dense f16 MFMA peak "
                   "(the pipe both kernels run on).  lin_z (22.9 % of the algorithmic FLOP at NS=1) is applied to the feature grid "
                   "once per scene (per-texel tables, re-done inside every timed step) and reaches the samples by bilinear lookup; "
                   "executed_mfma_tflops = the GEMM work really issued per sample x MFMAs per product (3 for f16x3)"}
    if derived:
        blk["pmc_derived"] = derived
    return blk


def eager_gpu_baseline(scene, mlps, rays, dev, n=16384, calls=3, eval_batch_size=None):
    """The north_star's "reference single-GPU" comparison point: the same eager PyTorch fp32 code path (the oracle
    restatement of the reference, with F.grid_sample like the reference) on THIS GPU through PyTorch-ROCm.
    eval_batch_size=None: one model call per pass (no chunk loop).  An integer: the reference's EXECUTION SHAPE --
    eval/eval.py:264 splits the rays into ray_batch_size = 50 000 per render_par call (util/args.py:19) and eval/eval.py:137
    hands the same number to the renderer as eval_batch_size, so composite() walks the points in chunks of 50 000 per model
    call (nerf.py:190-216): 64 model calls for the coarse pass of such a ray batch, 192 for the fine pass.
    A reported baseline only -- never part of the product path."""
    from oracle import pnr_oracle as O
    from testdata import synthetic
    O.USE_GRID_SAMPLE = True
    sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
    ms = [{k: v.to(dev) for k, v in m.items()} for m in mlps]
    r = rays[:n]
    noise = {k: v.to(dev) for k, v in synthetic.make_noise(r.shape[0], 64, 128, 16, seed=7).items()}
    try:
        with torch.no_grad():
            # warm-up at the SAME shapes (GEMM heuristics, caching-allocator growth), then steady-state calls are timed
            O.render(sc, ms[0], ms[1], r[None], noise, 64, 128, 16, white_bkgd=True, eval_batch_size=eval_batch_size)
            torch.cuda.synchronize()
            dts = []
            for _ in range(calls):
                t0 = time.perf_counter()
                O.render(sc, ms[0], ms[1], r[None], noise, 64, 128, 16, white_bkgd=True, eval_batch_size=eval_batch_size)
                torch.cuda.synchronize()
                dts.append(time.perf_counter() - t0)
    finally:
        O.USE_GRID_SAMPLE = False
    dt = sum(dts) / len(dts)
    shape = ("one model call per pass (no eval_batch_size chunk loop)" if eval_batch_size is None else
             "the reference's execution shape: ray batch %d (eval/eval.py:264, util/args.py:19), model calls of %d points "
             "(eval/eval.py:137 -> nerf.py:190-216)" % (r.shape[0], eval_batch_size))
    return {"value": r.shape[0] / dt, "unit": "rays/s", "kind": "port (oracle restatement, torch fp32 eager on the same MI355X)",
            "shape": shape,
            "sample": "%d rays per call, mean of %d steady-state calls (after a same-shape warm-up): %s s" % (
                r.shape[0], calls, ", ".join("%.3f" % d for d in dts))}


def encode_timing(dev, n_img=16, reps=5):
    """encode time reported separately.  SpatialEncoder (ResNet-34 trunk in PyTorch-ROCm, random init, eval
    mode) + pnr_pyramid_to_latent on sn64-shaped inputs (64x64 images, use_first_pool=False -> 512 x 32 x 32 grid)."""
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.util.conf import default_model_conf
    conf = default_model_conf()
    net = make_model(conf, precision="f16x3").to(dev).eval()
    out = {}
    with torch.no_grad():
        for n in (1, n_img):
            img = torch.rand(n, 3, 64, 64, device=dev) * 2 - 1
            net.encoder(img)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                lat = net.encoder(img)
            torch.cuda.synchronize()
            out["%d_images_ms" % n] = (time.perf_counter() - t0) / reps * 1e3
            out["latent_shape"] = list(lat.shape)
    out["what"] = ("SpatialEncoder.forward on (n,3,64,64): plain-torch ResNet-34 trunk (PyTorch-ROCm, fp32, eval) + "
                   "pnr_pyramid_to_latent; excluded from `value` (SURVEY 8d)")
    return out


def extra_render_config(dev, scene_name, n_img, n_oracle=128, n_f32=8192, steps=3, precisions=("f16x3", "f16")):
    """One of BASELINE configs[2] (srn_car 128x128, 2 views) / configs[3] on one GPU (DTU 400x300, 3 views, 176 MiB
    grid) / the reference's 9-view DTU evaluation (README.md:202: dtu_9v, 553 MB grid, 1.66 GB of tables per network;
    11.51 GFLOP/ray = 256 x (9 x 4.7616 + 2.1012) MFLOP): rays/s through render_par(rays) at 64+128 for the fp32-class path and the f16 path, PSNR of a ray sample spread
    over the whole image (border pixels included) vs the CPU oracle and vs the exact-fp32 HIP path.  Untimed extras."""
    from oracle import pnr_oracle as O
    from testdata import synthetic
    scene, meta, net, renderer, mlps = build(dev, "f16x3", scene_name)
    NS = scene["NS"]
    rays1 = synthetic.target_rays(meta).reshape(-1, 8)
    rays = rays1.repeat(n_img, 1).contiguous().to(dev)
    R = rays.shape[0]
    render_par = renderer.bind_parallel(net, None, simple_output=True).eval()
    out = {"workload": "%s %dx%d, %d source views, grid %s, 64+128, %d rays per call" % (
        scene_name, meta["W"], meta["H"], NS, "x".join(str(v) for v in scene["latent"].shape), R)}
    g = torch.Generator().manual_seed(3)
    n_pix = rays1.shape[0]
    idx = torch.randperm(n_pix, generator=g)[:n_f32]
    W = meta["W"]
    border = torch.cat([torch.arange(0, W, max(W // 16, 1)), n_pix - 1 - torch.arange(0, W, max(W // 16, 1))])  # first / last rows
    idx[:border.numel()] = border
    rs = rays1[idx]
    noise = synthetic.make_noise(rs.shape[0], 64, 128, 16, seed=5)
    nz = {k: v.to(dev) for k, v in noise.items()}
    span = float(meta["z_far"] - meta["z_near"])
    with torch.no_grad():
        net.precision = "f32"
        exact = renderer(net, rs.to(dev)[None], _noise=nz)
        no = min(n_oracle, rs.shape[0])
        ref = O.render(scene, mlps[0], mlps[1], rs[None, :no], {k: v[:no] for k, v in noise.items()}, 64, 128, 16,
                       white_bkgd=meta["white_bkgd"])
        for prec in precisions:
            net.precision = prec
            render_par(rays[None])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                net._tables.clear()  # a freshly encoded scene per step: the lin_z fold is inside the time
                render_par(rays[None])
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            fast = renderer(net, rs.to(dev)[None], _noise=nz)
            alg = R / dt * flop_per_ray(NS) / 1e12
            # the per-scene fold of lin_z into the grid on its own (both networks; inside ms_per_call above): the per-rank cost of a
            # sharded render that does not shrink with the rank count
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                net._tables.clear()
                net.tables(True), net.tables(False)
            e1.record()
            torch.cuda.synchronize()
            fold_ms = e0.elapsed_time(e1) / 3
            out[prec] = {"rays_per_s": R / dt, "ms_per_call": dt * 1e3, "fold_ms_both_networks": fold_ms, "algorithmic_tflops": alg,
                         "frac_of_f16_mfma_peak": alg / PEAK_TFLOPS,
                         "executed_mfma_tflops": alg * executed_fraction(NS, True) * MFMAS_PER_PRODUCT[prec],
                         "psnr_db_vs_cpu_oracle": O.psnr(fast.fine.rgb[0, :no].cpu(), ref["fine"]["rgb"][0]), "oracle_rays": no,
                         "psnr_db_vs_f32_hip": O.psnr(fast.fine.rgb.cpu(), exact.fine.rgb.cpu()), "f32_rays": int(rs.shape[0]),
                         "rgb_max_abs_err_vs_f32_hip_coarse": float((fast.coarse.rgb - exact.coarse.rgb).abs().max()),
                         "depth_abs_err_p99_over_span_vs_f32_hip": float(torch.quantile((fast.fine.depth - exact.fine.depth).abs().flatten(), 0.99)) / span}
    del net, renderer
    torch.cuda.empty_cache()
    return out


def extra_eval_object_loop(dev, n_views=24, n_obj=4):
    """SURVEY 8f rows f1-f4 measured as the LOOP they replace.  One sn64 object = encode one 64x64 source view
    + render 24 target views at 64+128 + the evaluation epilogue, in two forms over the same network and the same HIP renderer:
      (i)  reference-shaped, eval/eval.py:247-290,327-329: util.gen_rays on the HOST -> .to(device) -> render_par per 50 000-ray
           chunk -> .cpu() per chunk -> torch.cat / clamp / numpy uint8 / numpy PSNR per view;
      (ii) pnr_render_views (poses -> pixels in one C call, rays never materialised) + pnr_eval_epilogue (clamp, uint8, normalised
           depth, per-view PSNR on the device), ONE device-to-host copy at the end.
    Both draw their jitter in-kernel; the timed (i) is the plain loop (one Philox key per chunk), so its images differ from (ii)'s by
    the draws.  The untimed identity check runs (i) with the chunk's place in the whole ray set handed to the renderer
    (ray_id_offset / ray_id_stride / one key: what the multi-device wrapper does): then the uint8 images must be IDENTICAL."""
    import numpy as np
    from pixelnerf_amd import ops, util
    from testdata import synthetic
    scene, meta, net, renderer, mlps = build(dev, "f16x3", "sn64")
    W, H, P = meta["W"], meta["H"], meta["W"] * meta["H"]
    z_near, z_far, focal_xy, c_xy = meta["z_near"], meta["z_far"], meta["focal"], meta["c"]
    rs = np.random.RandomState(7)
    images = torch.from_numpy(rs.uniform(-1, 1, (n_obj, 1 + n_views, 3, H, W)).astype(np.float32))  # view 0 = source, 1.. = ground truth
    src_pose = synthetic.pose_spherical(30.0, -20.0, meta["radius"])[None]
    tgt_poses = torch.stack([synthetic.pose_spherical(45.0 + 13.0 * i, -20.0, meta["radius"]) for i in range(n_views)])
    focal = torch.tensor(focal_xy[0], dtype=torch.float32)[None]
    c = torch.tensor(c_xy, dtype=torch.float32)[None]
    render_par = renderer.bind_parallel(net, None, simple_output=True).eval()
    ray_batch = 50000  # eval/eval.py:135-137 (eval_batch_size = ray_batch_size = 50 000 in the shipped confs)

    def ref_shaped(o, place_chunks=False, key=None, device_rays=False, encode=True):
        if device_rays:  # (identity check only) the gen_rays KERNEL: the same bits pnr_render_views regenerates per pixel
            all_rays = util.gen_rays(tgt_poses.to(dev), W, H, focal, z_near, z_far, c=c).reshape(-1, 8)
        else:
            all_rays = util.gen_rays(tgt_poses, W, H, focal, z_near, z_far, c=c).reshape(-1, 8).to(device=dev)   # host rays + upload
        rays_spl = torch.split(all_rays, ray_batch, dim=0)
        if encode:
            net.encode(images[o, :1].to(device=dev).unsqueeze(0), src_pose.to(dev).unsqueeze(0), focal.to(dev), c=c.to(dev))
        all_rgb, all_depth, lo = [], [], 0
        for rays in rays_spl:
            if place_chunks:
                renderer.ray_id_offset, renderer.ray_id_stride, renderer._seed_override = lo, all_rays.shape[0], key
            rgb, depth = render_par(rays[None])
            all_rgb.append(rgb[0].cpu())
            all_depth.append(depth[0].cpu())
            lo += rays.shape[0]
        renderer.ray_id_offset, renderer.ray_id_stride, renderer._seed_override = 0, 0, None
        all_rgb, all_depth = torch.cat(all_rgb, dim=0), torch.cat(all_depth, dim=0)
        depth_n = ((all_depth - z_near) / (z_far - z_near)).reshape(n_views, H, W).numpy()
        rgb01 = torch.clamp(all_rgb.reshape(n_views, H, W, 3), 0.0, 1.0).numpy()
        u8 = (rgb01 * 255).astype(np.uint8)
        gt = (images[o, 1:] * 0.5 + 0.5).permute(0, 2, 3, 1).contiguous().numpy()
        psnr = [-10.0 * np.log10(np.mean((rgb01[v].astype(np.float64) - gt[v]) ** 2)) for v in range(n_views)]  # compare_psnr, data_range 1
        return u8, depth_n, np.array(psnr)

    def hip_side(o, key=None):
        net.encode(images[o, :1].to(device=dev).unsqueeze(0), src_pose.to(dev).unsqueeze(0), focal.to(dev), c=c.to(dev))
        pk_c, pk_f = net.packed(True), net.packed(False)
        guarded = net._guard_begin()  # the fp16-range guard, as every render_par call of form (i) runs it
        try:
            out = ops.render_views(net.scene(), pk_c, pk_f, tgt_poses.to(dev), W, H, focal_xy, z_near, z_far, 64, 128, 16, None, c=c_xy,
                                   white_bkgd=meta["white_bkgd"], tables=(net.tables(True), net.tables(False)),
                                   seed=renderer._next_seed(dev) if key is None else key)
        finally:
            if guarded:
                net._guard_end()
        gt = (images[o, 1:].to(dev) * 0.5 + 0.5).permute(0, 2, 3, 1).reshape(n_views, P, 3).contiguous()
        ep = ops.eval_epilogue(out["fine"]["rgb"].reshape(n_views, P, 3), out["fine"]["depth"].reshape(n_views, P), z_near, z_far, gt)
        packed = torch.cat([ep["rgb_u8"].reshape(n_views, -1).float(), ep["depth_norm"], ep["psnr"].float()[:, None]], dim=1).cpu()  # ONE D2H
        u8 = packed[:, :P * 3].to(torch.uint8).reshape(n_views, H, W, 3).numpy()
        return u8, packed[:, P * 3:P * 4].reshape(n_views, H, W).numpy(), packed[:, -1].double().numpy()

    res = {"workload": "one sn64 object = encode 1 source view (ResNet-34, 64x64) + %d target views of 64x64 at 64+128 + eval epilogue; "
                       "%d objects per timing; precision f16x3" % (n_views, n_obj)}
    with torch.no_grad():
        ref_shaped(0), hip_side(0)  # warm: packs, folds, encoder graph
        torch.cuda.synchronize()
        for name, fn in (("reference_shaped_loop", ref_shaped), ("render_views_plus_epilogue", hip_side)):
            t0 = time.perf_counter()
            for o in range(n_obj):
                last = fn(o)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_obj
            res[name] = {"ms_per_object": dt * 1e3, "images_per_s": n_views / dt, "rays_per_s": n_views * P / dt,
                         "mean_psnr_db_vs_random_gt": float(np.mean(last[2]))}
        res["speedup"] = res["reference_shaped_loop"]["ms_per_object"] / res["render_views_plus_epilogue"]["ms_per_object"]
        key = 0x1234567
        b = hip_side(1, key=key)
        for name, dr in (("same_draws_check", True), ("same_draws_check_host_rays", False)):
            # on the scene hip_side(1) encoded: the torch / MIOpen ResNet-34 trunk is not reproducible from call to call on these boxes
            # (layer2 / layer3 outputs of the SAME image differ by 4e-6 / 4e-5, tools/experiments/enc_diag.py: its 8x8 and 4x4 maps run
            # split-K convolutions); a second encode would put ulp-level differences under both forms and a white pixel at 1 - 1 ulp
            # truncates to 254 (sessions r06_s20 / s24 / s30: ~12 000 of 294 912 values off by one; r06_s9 / s27: none)
            a = ref_shaped(1, place_chunks=True, key=key, device_rays=dr, encode=False)
            diff = np.abs(a[0].astype(np.int32) - b[0].astype(np.int32))
            res[name] = {"u8_images_identical": bool(diff.max() == 0), "u8_max_abs_diff": int(diff.max()),
                         "u8_values_differing": int((diff > 0).sum()), "u8_values": int(diff.size),
                         "depth_norm_max_abs_diff": float(np.abs(a[1] - b[1]).max()),
                         "psnr_db_max_abs_diff": float(np.abs(a[2] - b[2]).max())}
        res["same_draws_check"]["what"] = ("on ONE encoded scene (the torch trunk is not bit-reproducible between two encodes), form (i) with rays from the gen_rays KERNEL and every chunk placed in the whole ray set: the bits of form (ii).  "
                                           "`same_draws_check_host_rays`: the reference's host-side util.gen_rays differs from the kernel in the last bit of "
                                           "some ray directions; a uint8 value changes where a colour sits on a truncation boundary")
        mse = np.mean((ref_shaped(2)[0].astype(np.float64) - hip_side(2)[0].astype(np.float64)) ** 2)
        res["psnr_db_between_u8_images_independent_draws"] = float(10 * np.log10(255.0 ** 2 / max(mse, 1e-12)))
    del net, renderer
    torch.cuda.empty_cache()
    return res


def extra_train_step(dev, prec, scene_name="train", steps=40, warmup=8, with_graph=True):
    """BASELINE configs[4]: sn64 training step, 4 objects x 128 rays, 64 coarse + 32 fine (16 depth), ResnetFC d=512,
    forward + backward (+ Adam) through NeRFRenderer/_RenderWrapper in train mode (train/train.py:199-215).
    scene_name "train_mv": the same step on a 2-object x 2-source-view scene (multi-view pooling in forward and backward);
    "dtu": 1 object x 3 views on the full DTU grid; "dtu_train4": the reference's DTU training batch (README.md:204,253: 4 objects x
    3 views x 128 rays; twelve 150 x 200 grids)."""
    from pixelnerf_amd.model import make_model
    from pixelnerf_amd.render import NeRFRenderer
    from pixelnerf_amd.util import DotMap
    from pixelnerf_amd.util.conf import default_model_conf
    from testdata import synthetic
    big = scene_name == "dtu_train4"  # 12 full-size DTU grids: drawn on the device (timing only)
    scene, meta = synthetic.make_scene(scene_name, with_latent=not big)
    SB, NS = scene["SB"], scene["NS"]
    if big:
        scene["latent"] = torch.randn((SB * NS, 512, meta["Hl"], meta["Wl"]), device=dev, generator=torch.Generator(device=dev).manual_seed(2)) * 0.5
    rays = synthetic.target_rays(meta, n_rays=128).to(dev)  # (SB,128,8)
    gt = torch.rand(SB, 128, 3, device=dev)
    net = make_model(default_model_conf(), precision=prec).to(dev).train()
    net.mlp_coarse.load_state_dict(synthetic.make_mlp_params(11))
    net.mlp_fine.load_state_dict(synthetic.make_mlp_params(12))
    lat = scene["latent"].to(dev).clone().requires_grad_(True)
    net.encoder.latent = lat
    ls = torch.tensor([float(lat.shape[-1]), float(lat.shape[-2])], device=dev)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses, net.image_shape = scene["poses"].to(dev), scene["image_shape"].to(dev)
    net.focal, net.c = scene["focal"].to(dev), scene["c"].to(dev)
    net.num_objs, net.num_views_per_obj = SB, NS
    rend = NeRFRenderer(n_coarse=64, n_fine=32, n_fine_depth=16, white_bkgd=True).to(dev)
    render_par = rend.bind_parallel(net, None, simple_output=False).train()
    # the reference's optimizer (train/trainer.py: torch.optim.Adam); `fused=True` is PyTorch's single-kernel form of the
    # same update -- the optimizer is outside the hot path, so it is taken in its cheapest stock form
    try:
        opt = torch.optim.Adam(list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters()), lr=1e-4, fused=True)
    except Exception:
        opt = torch.optim.Adam(list(net.mlp_coarse.parameters()) + list(net.mlp_fine.parameters()), lr=1e-4)

    def step():
        rd = DotMap(render_par(rays, want_weights=True))
        loss = ((rd.coarse.rgb - gt) ** 2).mean() + ((rd.fine.rgb - gt) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        lat.grad = None
        loss.backward()
        opt.step()
        return loss

    loss_first = float(step().item())
    for _ in range(warmup - 1):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    n_rays = SB * 128
    flop_ray = 3.0 * flop_per_ray(NS, 64, 32)  # fwd + dX + dW
    out = {"workload": "%s: %d objects x 128 rays, %d source view(s), 64+32 (16 depth) samples, fwd+bwd+Adam, grads to both "
                       "ResnetFCs and encoder.latent" % (scene_name, SB, NS), "precision": prec,
           "ms_per_step": dt * 1e3, "steps_per_s": 1.0 / dt, "rays_per_s": n_rays / dt,
           "algorithmic_tflops": n_rays / dt * flop_ray / 1e12, "frac_of_f16_mfma_peak": n_rays / dt * flop_ray / 1e12 / PEAK_TFLOPS,
           "loss_first_step": loss_first, "loss": float(loss.item()), "steps": steps,
           "launch_mode": "eager launches (one Python-sequenced HIP launch per kernel)"}
    if prec == "f16x3":
        out["form"] = ("fused: split-operand forward kernel in its training instantiation, one launch for the transposed products, "
                       "one batched split-operand weight-gradient launch (pnr_eval_ray_samples_split_train / pnr_mlp_backward_split)")
    if with_graph:
        # the same step captured ONCE into a HIP graph (torch.cuda.CUDAGraph: forward, backward and a capturable Adam) and
        # replayed: no per-kernel launch latency, no Python between the ~35 kernels.  Run as a CHILD process
        # (tools/gpu_train_graph.py): a failure inside graph capture must never take this benchmark line down.
        import subprocess
        try:
            child = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_train_graph.py"), "step", "--prec", prec],
                                   capture_output=True, text=True, timeout=300)
            line = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
            if child.returncode == 0 and line:
                r = json.loads(line[-1])["step"]
                out["hip_graph"] = {"ms_per_step": r["graph_ms"], "steps_per_s": 1e3 / r["graph_ms"], "rays_per_s": 512e3 / r["graph_ms"],
                                    "algorithmic_tflops": 512e3 / r["graph_ms"] * 3.29e9 / 1e12, "eager_ms_per_step_same_process": r["eager_ms"],
                                    "loss": r["loss"]}
            else:
                out["hip_graph"] = {"error": "child rc=%d: %s" % (child.returncode, child.stderr[-300:])}
        except Exception as e:
            out["hip_graph"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    del net, rend
    torch.cuda.empty_cache()
    return out


def train_step_eager_torch(dev, steps=4):
    """BASELINE configs[4] in eager PyTorch-ROCm fp32 autograd on THIS GPU: the oracle restatement of the reference (F.grid_sample
    like the reference) -- forward, loss (train/train.py:199-215), backward, Adam.  A reported baseline, never the product path."""
    from oracle import pnr_oracle as O
    from testdata import synthetic
    O.USE_GRID_SAMPLE = True
    try:
        scene, meta = synthetic.make_scene("train")
        sc = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
        sc["latent"] = sc["latent"].clone().requires_grad_(True)
        pc = {k: v.to(dev).requires_grad_(True) for k, v in synthetic.make_mlp_params(11).items()}
        pf = {k: v.to(dev).requires_grad_(True) for k, v in synthetic.make_mlp_params(12).items()}
        rays = synthetic.target_rays(meta, n_rays=128).to(dev)
        gt = torch.rand(scene["SB"], 128, 3, device=dev)
        noise = {k: v.to(dev) for k, v in synthetic.make_noise(scene["SB"] * 128, 64, 32, 16).items()}
        opt = torch.optim.Adam(list(pc.values()) + list(pf.values()), lr=1e-4)

        def step():
            out = O.render(sc, pc, pf, rays, noise, 64, 32, 16, white_bkgd=True)
            loss = ((out["coarse"]["rgb"] - gt) ** 2).mean() + ((out["fine"]["rgb"] - gt) ** 2).mean()
            opt.zero_grad(set_to_none=True)
            sc["latent"].grad = None
            loss.backward()
            opt.step()

        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    finally:
        O.USE_GRID_SAMPLE = False
    return {"ms_per_step": dt * 1e3, "steps": steps, "kind": "port (oracle restatement, torch fp32 eager autograd + Adam on the same MI355X)",
            "workload": "train: 4 objects x 128 rays, 64+32 (16 depth) samples"}


def train_step_unfused_twin(dev):
    """the fp32-class training step in its GEMM-per-layer form (one split-operand GEMM launch per product; the A/B twin of the
    fused default and the form the round started from)"""
    from pixelnerf_amd import autograd
    saved, autograd.FUSED_SPLIT_TRAINING = autograd.FUSED_SPLIT_TRAINING, False
    try:
        r = extra_train_step(dev, "f16x3", steps=6, warmup=2, with_graph=False)
    finally:
        autograd.FUSED_SPLIT_TRAINING = saved
    r["form"] = "one split-operand GEMM launch per linear (autograd.FUSED_SPLIT_TRAINING = False)"
    return r


def self_launch(n):
    """Re-run this script under torch.distributed.run with n ranks on this node (replaces the reference's
    single-process DataParallel seam, src/render/nerf.py:367-371)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_region(step, fence, steps, warmup):
    """W untimed warm-ups, then exactly K steps bracketed by fence() (synchronize [+ barrier + synchronize]); HIP events of the
    network-kernel launches on their stream are collected over the same K steps.  -> (elapsed s, kernel ms, launches)"""
    from pixelnerf_amd import ops
    for _ in range(warmup):
        step()
    fence()
    ops.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kern_ms, n_launch = ops.profile_read()
    ops.profile_enable(False)
    return elapsed, kern_ms, n_launch


def strong_dtu(dev, prec, world, rank, steps, warmup, bcast, fence, dist_on=None, layout="nhwc"):
    """BASELINE configs[3]: ONE DTU 400x300 image (120 000 rays, 3 source views, 176 MiB grid) sharded contiguously over
    the ranks.  step = [the single feature-grid broadcast from rank 0] + render of this rank's rays + [gather of
    (rgb | depth) to rank 0]; returns the rank-local timings (the caller max-reduces `elapsed`)."""
    import torch.distributed as dist
    from pixelnerf_amd.dist import broadcast_encoded, shard_bounds
    from testdata import synthetic
    scene, meta, net, renderer, _ = build(dev, prec, "dtu")
    lat_shape = tuple(scene["latent"].shape)
    rays_all = synthetic.target_rays(meta).reshape(-1, 8)
    R = rays_all.shape[0]
    lo, hi = shard_bounds(R, rank, world)
    rays = rays_all[lo:hi].contiguous().to(dev)
    render_par = renderer.bind_parallel(net, None, simple_output=True).eval()
    renderer.ray_id_offset, renderer.ray_id_stride = lo, R  # the sharded image equals the unsharded one
    tb, tg = [0.0], [0.0]
    dist_on = world > 1 if dist_on is None else dist_on

    def step():
        net._tables.clear()
        if dist_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            broadcast_encoded(net, src=0, latent_shape=lat_shape, algo=bcast, layout=layout)
            e1.record()
        with torch.no_grad():
            rgb, depth = render_par(rays[None])
        if dist_on:
            e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out = torch.cat([rgb[0], depth[0].unsqueeze(-1)], dim=-1)  # 16 B/ray
            sizes = [shard_bounds(R, r, world)[1] - shard_bounds(R, r, world)[0] for r in range(world)]
            if out.shape[0] < max(sizes):
                out = torch.cat([out, out.new_zeros(max(sizes) - out.shape[0], 4)])
            bufs = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
            e2.record()
            dist.gather(out, bufs, dst=0)
            e3.record()
            evs.append((e0, e1, e2, e3))
        return rgb

    evs = []
    torch.manual_seed(4321)
    elapsed, kern_ms, n_launch = timed_region(step, fence, steps, warmup)
    for e0, e1, e2, e3 in evs[-steps:]:
        tb[0] += e0.elapsed_time(e1)
        tg[0] += e2.elapsed_time(e3)
    NS = scene["NS"]
    res = {"elapsed": elapsed, "kern_ms": kern_ms, "n_launch": n_launch, "R": R, "rays_this_rank": hi - lo, "NS": NS,
           "bcast_ms": tb[0] / steps, "gather_ms": tg[0] / steps, "grid_bytes": int(scene["latent"].numel() * 4)}
    del net, renderer
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prec", default="f16x3", choices=["f16x3", "f16", "bf16"],
                    help="precision of the timed headline (default: the fp32-class path, the reference's arithmetic class)")
    ap.add_argument("--rays", type=int, default=65536, help="rays per GPU per step (sn64 workload)")
    ap.add_argument("--workload", default="sn64", choices=["sn64", "dtu"],
                    help="sn64: BASELINE configs[1], weak scaling (default).  dtu: configs[3], ONE 120 000-ray image sharded over the ranks (strong)")
    ap.add_argument("--bcast", default="tree", choices=["tree", "flat"],
                    help="feature-grid broadcast: RCCL's broadcast, or a flat 1->(N-1) fan-out of point-to-point sends (one xGMI link each)")
    ap.add_argument("--cpu-rays", type=int, default=0, help="CPU-baseline sample size (0 = auto, ~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true")
    ap.add_argument("--no-f32-check", action="store_true")
    ap.add_argument("--no-peer", action="store_true", help="skip the second timed region (the f16 peer block)")
    ap.add_argument("--no-latency", action="store_true", help="skip the 4096-ray latency / encode sections (profiling runs: every "
                    "network-kernel launch of the process then has the timed region's shape)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (gloo: functional test on one GPU)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with ONE rank: still create the process group (RCCL communicator) and run every step through the "
                         "distributed code path (grid broadcast + render + gather; `comm` block in the JSON)")
    ap.add_argument("--bcast-layout", default="nhwc", choices=["nhwc", "nchw"],
                    help="layout the feature grid travels in: channel-last (what the kernels read; receivers skip the transpose) or NCHW")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed BASELINE configs 3/4/5 section (extra.configs)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure `roofline.traffic` "
                    "in this run (profiling / A/B invocations; the committed profile is reported instead)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, RCCL), exactly the
        # command the driver would have used; rank 0 of the child job prints the single JSON line
        return self_launch(args.gpus)
    # stdout carries the ONE JSON line and nothing else: RCCL prints its version banner to the C-level stdout of every process
    # that creates a communicator (seen with --force-dist: five lines behind the JSON).  File descriptor 1 is pointed at stderr
    # for the rest of the run; the line goes out through a duplicate of the original descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    local_dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    import torch.distributed as dist
    dist_on = world > 1 or args.force_dist
    if dist_on and world == 1 and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl" and torch.cuda.device_count() < world:
            raise SystemExit("bench.py --gpus %d: RCCL needs one device per rank and this node has %d (use --backend gloo "
                             "for a functional run on fewer devices)" % (world, torch.cuda.device_count()))
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from pixelnerf_amd import _lib
    from testdata import synthetic
    from pixelnerf_amd.dist import broadcast_encoded

    _lib.ensure_built()  # normally a no-op: the .so built by __graft_entry__.build() travels with the tree

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    def max_over_ranks(x):
        if not dist_on:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    strong = args.workload == "dtu"
    res = None
    if strong:
        # ---- BASELINE configs[3] as the headline: one DTU image over N ranks
        st = strong_dtu(dev, args.prec, world, rank, args.steps, args.warmup, args.bcast, fence, dist_on, args.bcast_layout)
        elapsed = max_over_ranks(st["elapsed"])
        if rank == 0:
            rays_per_s = st["R"] * args.steps / elapsed
            res = {"metric": "rays/sec (64 coarse + 128 fine samples) at matched PSNR vs reference",
                   "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
                   "vs_baseline": None, "dtype": args.prec, "dtype_note": DTYPE_NOTE[args.prec], "data": "synthetic",
                   "config": {"workload": "DTU 400x300, 3 input views, 64+128 samples, ONE full-res image (120000 rays) sharded "
                                          "contiguously over %d rank(s) (BASELINE configs[3]); synthetic 3x512x150x200 feature grid "
                                          "(176 MiB), random-init ResnetFC coarse+fine; step = grid broadcast + render + gather" % world,
                              "rays_per_image": st["R"], "rays_rank0": st["rays_this_rank"], "source_views": st["NS"],
                              "rccl_ranks": world, "backend": args.backend, "bcast_algo": args.bcast},
                   "comm": {"bcast_ms_rank0": st["bcast_ms"], "gather_ms_rank0": st["gather_ms"], "grid_bytes": st["grid_bytes"]},
                   "roofline": roofline_block(args.prec, st["rays_this_rank"], args.steps, st["kern_ms"], st["n_launch"], st["NS"], True,
                                              elapsed, False)}
        if res is not None:
            emit(res)
        if dist_on:
            dist.destroy_process_group()
        return 0

    # ---- BASELINE configs[1]: sn64, weak scaling
    scene, meta, net, renderer, mlps = build(dev, args.prec)
    lat_shape = tuple(scene["latent"].shape)
    R = args.rays
    rays = make_rays(meta, R, rank).to(dev)
    render_par = renderer.bind_parallel(net, None, simple_output=True).eval()
    comm_ev = []

    def step():
        # every step stands for a freshly encoded object: the per-scene folding of lin_z into the feature grid
        # (PixelNeRFNet.tables -> pnr_fold_latent[_f32], both networks) is redone INSIDE the timed step
        net._tables.clear()
        if dist_on:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            broadcast_encoded(net, src=0, latent_shape=lat_shape, algo=args.bcast, layout=args.bcast_layout)  # THE single feature-grid broadcast (2 MiB for sn64)
            e1.record()
        with torch.no_grad():
            rgb, depth = render_par(rays[None])
        if dist_on:
            out = torch.cat([rgb[0], depth[0].unsqueeze(-1)], dim=-1)  # 16 B/ray
            bufs = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
            e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e2.record()
            dist.gather(out, bufs, dst=0)
            e3.record()
            comm_ev.append((e0, e1, e2, e3))
        return rgb

    torch.manual_seed(1234 + rank)
    net.precision = args.prec
    elapsed, kern_ms, n_launch = timed_region(step, fence, args.steps, args.warmup)
    elapsed = max_over_ranks(elapsed)
    comm = None
    if dist_on:
        ev = comm_ev[-args.steps:]
        comm = {"bcast_ms_rank0": sum(a.elapsed_time(b) for a, b, _, _ in ev) / len(ev),
                "gather_ms_rank0": sum(c.elapsed_time(d) for _, _, c, d in ev) / len(ev),
                "grid_bytes": int(scene["latent"].numel() * 4), "bcast_algo": args.bcast, "bcast_layout": args.bcast_layout,
                "backend": args.backend, "ranks": world}

    peer = None
    if world == 1 and not args.no_peer:
        peer_prec = "f16" if args.prec == "f16x3" else "f16x3"
        net.precision = peer_prec
        torch.manual_seed(1234 + rank)
        pe, pk, pn = timed_region(step, fence, args.steps, args.warmup)
        net.precision = args.prec
        peer = (peer_prec, pe, pk, pn)

    NS = scene["NS"]
    default_shape = R == 65536
    if rank == 0:
        rays_per_s = world * R * args.steps / elapsed
        res = {
            "metric": "rays/sec (64 coarse + 128 fine samples) at matched PSNR vs reference",
            "value": rays_per_s, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.prec, "dtype_note": DTYPE_NOTE[args.prec], "data": "synthetic",
            "config": {"workload": "sn64 NMR 64x64, 1 input view, 64+128 samples (BASELINE configs[1]); "
                                   "%d rays (%d target views) per GPU per step; synthetic 1x512x32x32 feature grid, "
                                   "random-init ResnetFC coarse+fine (d_hidden 512, 5 blocks)" % (R, R // 4096),
                       "rays_per_gpu_per_step": R, "n_coarse": 64, "n_fine": 128, "n_fine_depth": 16,
                       "source_views": NS, "api": "NeRFRenderer.bind_parallel(net, simple_output=True)(rays)",
                       "precision": args.prec, "lin_z_folded_into_grid": True, "rccl_ranks": world, "backend": args.backend if dist_on else None},
            "roofline": roofline_block(args.prec, R, args.steps, kern_ms, n_launch, NS, True, elapsed, default_shape),
        }
        if comm:
            res["comm"] = comm
        if peer:
            pp, pe, pk, pn = peer
            res[pp + "_path"] = {
                "what": "the same workload, same steps / warm-up, timed in a second region right after the headline: " + DTYPE_NOTE[pp],
                "value": R * args.steps / pe, "unit": "rays/s", "ms_per_step": pe / args.steps * 1e3, "dtype": pp,
                "roofline": roofline_block(pp, R, args.steps, pk, pn, NS, True, pe, default_shape)}
    if world == 1 and rank == 0 and default_shape and not args.no_live_pmc:
        # roofline.traffic measured in THIS run (outside the timed region): two counter passes around a 2-step child
        live, why, live_extras = measure_traffic_live(args.prec)
        rf = res["roofline"]
        if live is not None:
            rf["traffic_committed_profile"] = rf["traffic"]
            rf["traffic"] = live
            rf["traffic_source"] = ("measured in this run: rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE, WRITE_SIZE; one pass each) around "
                                    "a 2-step child of this command, average over its fused-kernel launches")
            rf.update(live_extras)  # shader clock during the kernel / MFMA-busy share of this box (third pass)
        else:
            rf["traffic_live_error"] = why
    if world == 1 and not args.no_cpu_baseline:
        # bounded CPU sample of the same workload; also yields the matched-PSNR figure.  torch's intra-op pool does not
        # scale to every core of a many-socket host on 512-wide GEMMs: pick the thread count that is FASTEST on a 64-ray
        # probe (fair to the CPU), then time the real sample with it.  `cores` reports the threads actually used.
        from oracle import pnr_oracle as O
        n0 = 64
        rs = rays[:4096].cpu()
        noise0 = synthetic.make_noise(n0, 64, 128, 16, seed=99)
        ncpu = os.cpu_count() or 1
        cpu_baseline(scene, mlps, rs[:n0], noise0, threads=min(ncpu, 16))  # warm-up
        best = (0.0, 1)
        for th in sorted({min(ncpu, t) for t in (8, 16, 32, 64, 128, ncpu)}):
            rate_t, _, _ = cpu_baseline(scene, mlps, rs[:n0], noise0, threads=th)
            if rate_t > best[0]:
                best = (rate_t, th)
        rate0 = best[0]
        torch.set_num_threads(best[1])
        n = args.cpu_rays or int(min(4096, max(256, rate0 * 12.0)) // 64 * 64)
        noise = synthetic.make_noise(n, 64, 128, 16, seed=100)
        rate, dt, ref = cpu_baseline(scene, mlps, rs[:n], noise, threads=best[1])
        nzd = {k: v.to(dev) for k, v in noise.items()}
        with torch.no_grad():
            out = renderer(net, rs[:n].to(dev)[None], _noise=nzd)
        res["psnr_db"] = O.psnr(out.fine.rgb.cpu(), ref["fine"]["rgb"])
        res["rgb_max_abs_err_vs_cpu"] = float((out.fine.rgb.cpu() - ref["fine"]["rgb"]).abs().max())
        res["depth_abs_err_p99"] = float(torch.quantile((out.fine.depth.cpu() - ref["fine"]["depth"]).abs().flatten(), 0.99))
        if peer:
            net.precision = peer[0]
            with torch.no_grad():
                outp = renderer(net, rs[:n].to(dev)[None], _noise=nzd)
            net.precision = args.prec
            res[peer[0] + "_path"]["psnr_db"] = O.psnr(outp.fine.rgb.cpu(), ref["fine"]["rgb"])
        res["cpu_baseline"] = {"value": rate, "unit": "rays/s", "cores": best[1], "host_cpus": ncpu, "kind": "port",
                               "sample": "%d rays of the same workload (64+128, same weights/grid), %.1f s, oracle/pnr_oracle.py "
                                         "(torch CPU fp32 restatement of the reference, F.grid_sample as in encoder.py:96-109)" % (n, dt)}
        res["speedup_vs_cpu_baseline"] = res["value"] / rate
        # BASELINE configs[0] verbatim: 32 coarse samples, no fine pass, ONE call with all 4096 rays of a view, CPU
        noise1 = synthetic.make_noise(4096, 32, 0, 0, seed=102)
        cpu_baseline(scene, mlps, rs[:256], {k: v[:256] for k, v in noise1.items()}, threads=best[1], n_coarse=32, n_fine=0, n_fine_depth=0)
        rate1, dt1, ref1 = cpu_baseline(scene, mlps, rs, noise1, threads=best[1], n_coarse=32, n_fine=0, n_fine_depth=0)
        from pixelnerf_amd.render import NeRFRenderer
        rend1 = NeRFRenderer(n_coarse=32, n_fine=0, n_fine_depth=0, white_bkgd=True).to(dev).eval()
        with torch.no_grad():
            nz1 = {k: v.to(dev) for k, v in noise1.items()}
            rend1(net, rs.to(dev)[None], _noise=nz1)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                o1 = rend1(net, rs.to(dev)[None], _noise=nz1)
            torch.cuda.synchronize()
            dtg = (time.perf_counter() - t1) / 5
        res["cpu_baseline_config1"] = {
            "value": rate1, "unit": "rays/s", "cores": best[1], "host_cpus": ncpu, "kind": "port",
            "sample": "BASELINE configs[0]: sn64, 1 view, 32 coarse samples (no fine pass), ray_batch=4096 in ONE call on the CPU path, %.1f s" % dt1,
            "hip_same_call_ms": dtg * 1e3, "hip_same_call_rays_per_s": 4096 / dtg, "hip_precision": args.prec,
            "psnr_db_hip_vs_cpu": O.psnr(o1.coarse.rgb.cpu(), ref1["coarse"]["rgb"])}
    if world == 1 and not args.no_latency:
        # single-image latency (one 4096-ray call through render_par, fold included) next to the saturated rate
        lat_ms = {}
        with torch.no_grad():
            for prec in (args.prec,) + ((peer[0],) if peer else ()):
                net.precision = prec
                r1 = rays[None, :4096]
                render_par(r1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    net._tables.clear()
                    render_par(r1)
                torch.cuda.synchronize()
                lat_ms[prec] = (time.perf_counter() - t0) / 10 * 1e3
        net.precision = args.prec
        res["latency_4096_rays_ms"] = lat_ms[args.prec]
        res["latency_4096_rays_ms_by_precision"] = lat_ms
        try:
            res["encode_ms"] = encode_timing(dev)
        except Exception as e:
            res["encode_ms"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    if world == 1 and not args.no_f32_check:
        # full-size cross-check on the GPU, outside the timed region: all R rays of the step with the same noise through
        # the exact, unfused fp32-MFMA validation path (agrees with the reference to ~1e-6)
        from oracle import pnr_oracle as O
        noise = {k: v.to(dev) for k, v in synthetic.make_noise(R, 64, 128, 16, seed=101).items()}

        def render_at(prec):
            net.precision = prec
            with torch.no_grad():
                renderer(net, rays[None, :4096], _noise={k: v[:4096] for k, v in noise.items()})  # pack / fold / warm
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                o = renderer(net, rays[None], _noise=noise)
                torch.cuda.synchronize()
            return o, time.perf_counter() - t1

        exact, dt32 = render_at("f32")
        fast, _ = render_at(args.prec)
        res["psnr_db_full_size_vs_f32_hip"] = O.psnr(fast.fine.rgb.cpu(), exact.fine.rgb.cpu())
        res["rgb_max_abs_diff_full_size_vs_f32_hip_coarse"] = float((fast.coarse.rgb - exact.coarse.rgb).abs().max())
        if peer:
            pfast, _ = render_at(peer[0])
            res[peer[0] + "_path"]["psnr_db_full_size_vs_f32_hip"] = O.psnr(pfast.fine.rgb.cpu(), exact.fine.rgb.cpu())
        net.precision = args.prec
        res["f32_unfused_validation_path_rays_per_s"] = R / dt32
    if world == 1 and not args.no_eager_baseline:
        # the comparison point north_star names, in the shape the reference runs it (50 000-ray calls, 50 000-point model
        # calls); the unchunked 16 384-ray form earlier rounds quoted stays next to it under its own key
        res["torch_eager_gpu_baseline"] = eager_gpu_baseline(scene, mlps, rays, dev, n=50000, eval_batch_size=50000)
        res["torch_eager_gpu_baseline_unchunked_16384"] = eager_gpu_baseline(scene, mlps, rays, dev)
        res["speedup_vs_torch_eager_gpu"] = res["value"] / res["torch_eager_gpu_baseline"]["value"]
        res["speedup_vs_torch_eager_gpu_unchunked_16384"] = res["value"] / res["torch_eager_gpu_baseline_unchunked_16384"]["value"]
        # the un-suffixed keys changed meaning between BENCH_r03 (one 16 384-ray call) and BENCH_r04 (the reference's
        # execution shape).  Both forms carry an explicit name from schema 5 on; the un-suffixed pair keeps its r04 meaning.
        res["bench_schema"] = 5
        res["torch_eager_gpu_baseline_ref_shape"] = res["torch_eager_gpu_baseline"]
        res["speedup_vs_torch_eager_gpu_ref_shape"] = res["speedup_vs_torch_eager_gpu"]
        res["baseline_keys_note"] = ("schema >= 4 (BENCH_r04 on): torch_eager_gpu_baseline / speedup_vs_torch_eager_gpu = *_ref_shape "
                                     "(50 000-ray batches, 50 000-point model calls, eval/eval.py:137,264); schema <= 3 (BENCH_r01-r03): "
                                     "the same keys held what is now *_unchunked_16384")
        if peer:
            res[peer[0] + "_path"]["speedup_vs_torch_eager_gpu"] = res[peer[0] + "_path"]["value"] / res["torch_eager_gpu_baseline"]["value"]
    del net, renderer, render_par
    torch.cuda.empty_cache()
    if not args.no_extras:
        extra = {}
        if world == 1:
            # BASELINE configs[2..4] on this one GPU, outside the timed region
            for key, fn in (("train_step", lambda: extra_train_step(dev, "f16")),
                            ("train_step_multiview", lambda: extra_train_step(dev, "f16", "train_mv", steps=20, warmup=4, with_graph=False)),
                            ("train_step_fp32_class", lambda: extra_train_step(dev, "f16x3", steps=16, warmup=4, with_graph=True)),
                            ("train_step_fp32_class_multiview", lambda: extra_train_step(dev, "f16x3", "train_mv", steps=12, warmup=3, with_graph=False)),
                            ("train_step_fp32_class_dtu", lambda: extra_train_step(dev, "f16x3", "dtu", steps=12, warmup=3, with_graph=False)),
                            ("train_step_fp32_class_dtu_batch4", lambda: extra_train_step(dev, "f16x3", "dtu_train4", steps=8, warmup=3, with_graph=False)),
                            ("train_step_fp32_class_gemm_per_layer", lambda: train_step_unfused_twin(dev)),
                            ("train_step_torch_eager_gpu_baseline", lambda: train_step_eager_torch(dev)),
                            ("train_step_fp32_validation_path", lambda: extra_train_step(dev, "f32", steps=5, warmup=2, with_graph=False)),
                            ("srn_car", lambda: extra_render_config(dev, "srn_car", 4)),
                            ("dtu", lambda: extra_render_config(dev, "dtu", 1)),
                            ("dtu_9v", lambda: extra_render_config(dev, "dtu_9v", 1, n_oracle=32, n_f32=2048, steps=2, precisions=("f16x3",))),
                            ("eval_object_loop", lambda: extra_eval_object_loop(dev))):
                if key == "train_step_torch_eager_gpu_baseline" and args.no_eager_baseline:
                    continue
                try:
                    extra[key] = fn()
                except Exception as e:  # an extra must never take the headline line down with it
                    extra[key] = {"error": "%s: %s" % (type(e).__name__, e)}
            eager_ms = extra.get("train_step_torch_eager_gpu_baseline", {}).get("ms_per_step")
            if eager_ms:
                for key in ("train_step", "train_step_fp32_class"):
                    if "ms_per_step" in extra.get(key, {}):
                        extra[key]["speedup_vs_torch_eager_gpu"] = eager_ms / extra[key]["ms_per_step"]
            if rank == 0:
                res["extra"] = {"configs": extra}
        else:
            # BASELINE configs[3] in its strong-scaling form on the same N ranks (a few steps, after the headline)
            try:
                st = strong_dtu(dev, args.prec, world, rank, max(2, min(args.steps, 5)), 1, args.bcast, fence, dist_on, args.bcast_layout)
                st_elapsed = max_over_ranks(st["elapsed"])
                nst = max(2, min(args.steps, 5))
                if rank == 0:
                    res["extra"] = {"strong_dtu": {
                        "workload": "BASELINE configs[3]: ONE DTU 400x300 image (120000 rays, 3 views, 176 MiB grid) sharded over %d ranks; "
                                    "step = grid broadcast + render + gather" % world,
                        "scaling": "strong", "n_gpus": world, "steps": nst, "rays_per_s": st["R"] * nst / st_elapsed,
                        "ms_per_image": st_elapsed / nst * 1e3, "bcast_ms_rank0": st["bcast_ms"], "gather_ms_rank0": st["gather_ms"],
                        "grid_bytes": st["grid_bytes"], "bcast_algo": args.bcast, "dtype": args.prec}}
            except Exception as e:
                if rank == 0:
                    res["extra"] = {"strong_dtu": {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}}
    if rank == 0:
        emit(res)
    if dist_on:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
