#!/usr/bin/env python3
"""
Staged GPU diagnosis of the fused network kernel against the CPU oracle (run on the GPU box:
`python tools/gpu_debug.py > gpurun_out/debug.txt`).  Each stage enables one more group of
layers so that a wrong fragment map / packing permutation shows up at the stage that
introduces it.  Not a test (tests/test_hip_parity.py asserts); a microscope.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pnr_oracle as O  # noqa: E402
from pixelnerf_amd import ops  # noqa: E402
from testdata import synthetic  # noqa: E402


def staged_params(full, stage):
    p = {k: torch.zeros_like(v) for k, v in full.items()}
    keep = {
        0: ["lin_in."],
        1: ["lin_in.", "lin_z.0."],
        2: ["lin_in.", "lin_z.0.", "blocks.0."],
        3: ["lin_in.", "lin_z.", "blocks.0.", "blocks.1.", "blocks.2."],
        4: None,
    }[stage]
    for k in full:
        if keep is None or any(k.startswith(s) for s in keep) or k.startswith("lin_out."):
            p[k] = full[k].clone()
    return p


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for scene_name in ("sn64", "mv_mini", "dtu_mini"):
        scene, meta = synthetic.make_scene(scene_name)
        SB, NS = scene["SB"], scene["NS"]
        B = 200
        g = torch.Generator().manual_seed(5)
        xyz = (torch.rand(SB, B, 3, generator=g) * 2 - 1)
        vd = torch.randn(SB, B, 3, generator=g)
        vd = vd / vd.norm(dim=-1, keepdim=True)
        dscene = ops.make_scene(scene["latent"].to(dev), scene["poses"].to(dev), scene["focal"].to(dev),
                                scene["c"].to(dev), scene["image_shape"], NS)
        full = synthetic.make_mlp_params(11)
        for prec in ("f16", "bf16"):
            for stage in range(5):
                p = staged_params(full, stage)
                ref, hid = O.pixelnerf_forward(scene, p, xyz, vd, return_hidden=True)
                packed = ops.pack_mlp({k: v.to(dev) for k, v in p.items()}, prec)
                dump = torch.full((SB * B, 512), float("nan"), device=dev)
                ops.debug_set_x_dump(dump)
                out = ops.eval_points(dscene, packed, xyz.to(dev), vd.to(dev))
                torch.cuda.synchronize()
                ops.debug_set_x_dump(None)
                ex = (dump.cpu().reshape(SB, B, 512) - hid).abs()
                eo = (out.cpu() - ref).abs()
                print(f"{scene_name:9s} NS={NS} {prec:5s} stage {stage}: |x| max {hid.abs().max():8.3f}  "
                      f"x err max {ex.max():.3e} mean {ex.mean():.3e}   rgb err max {eo[..., :3].max():.3e}  "
                      f"sigma err max {eo[..., 3].max():.3e} (|sigma| max {ref[..., 3].max():.3f})", flush=True)
                if stage == 0 and ex.max() > 0.1:
                    # locate the damage: per-feature and per-point error pattern
                    bad_f = (ex.reshape(-1, 512).max(0)[0] > 0.1).nonzero().flatten()[:40].tolist()
                    bad_p = (ex.reshape(-1, 512).max(1)[0] > 0.1).nonzero().flatten()[:40].tolist()
                    print("   bad features:", bad_f)
                    print("   bad points  :", bad_p)
    print("done")


if __name__ == "__main__":
    main()
