"""
Host-side helpers mirroring the names the hot path and its callers use from the reference's
src/util/util.py.  Only what sits on or directly beside the path is here.
"""
import math

import torch

from .conf import Conf  # noqa: F401
from .dotmap import DotMap  # noqa: F401


def repeat_interleave(input, repeats, dim=0):
    """src/util/util.py:58-65 (expand+reshape replication along axis 0)."""
    output = input.unsqueeze(1).expand(-1, repeats, *input.shape[1:])
    return output.reshape(-1, *input.shape[1:])


def combine_interleaved(t, inner_dims=(1,), agg_type="average"):
    """src/util/util.py:461-471.  (The fused kernel does this reduction in registers; this
    function exists for callers that use it directly.)"""
    if len(inner_dims) == 1 and inner_dims[0] == 1:
        return t
    t = t.reshape(-1, *inner_dims, *t.shape[1:])
    if agg_type == "average":
        return torch.mean(t, dim=1)
    if agg_type == "max":
        return torch.max(t, dim=1)[0]
    raise NotImplementedError("Unsupported combine type " + agg_type)


def psnr(pred, target):
    """src/util/util.py:474-481."""
    mse = ((pred - target) ** 2).mean()
    return -10 * math.log10(mse)


def get_cuda(gpu_id):
    """src/util/util.py:193-199."""
    return torch.device("cuda:%d" % gpu_id) if torch.cuda.is_available() else torch.device("cpu")


def coord_from_blender(dtype=torch.float32, device="cpu"):
    """src/util/util.py:146-157."""
    return torch.tensor([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=dtype, device=device)


def coord_to_blender(dtype=torch.float32, device="cpu"):
    """src/util/util.py:160-171."""
    return torch.tensor([[1, 0, 0, 0], [0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=dtype, device=device)


def pose_spherical(theta, phi, radius):
    """src/util/util.py:309-323."""
    from ..synthetic import pose_spherical as _ps
    return _ps(theta, phi, radius)


def unproj_map(width, height, f, c=None, device="cpu"):
    """src/util/util.py:113-143 (host tensors)."""
    from ..synthetic import unproj_map as _um
    if torch.is_tensor(f):
        f = float(f) if f.numel() == 1 else (float(f.flatten()[0]), float(f.flatten()[1]))
    if torch.is_tensor(c):
        c = c.flatten().tolist()
    return _um(width, height, f, c=c).to(device)


def gen_rays(poses, width, height, focal, z_near, z_far, c=None, ndc=False):
    """src/util/util.py:238-276.  poses (B,4,4) camera-to-world -> rays (B,H,W,8).
    HIP tensors go through the gen_rays kernel (pnr_gen_rays); host tensors are input
    preparation exactly as in the reference and stay on the host."""
    if ndc:
        raise NotImplementedError("ndc rays are not used by any shipped config (util.py:250-259)")
    if torch.is_tensor(focal):
        focal = focal.flatten().tolist()
        focal = focal[0] if len(focal) == 1 else (focal[0], focal[1])
    if torch.is_tensor(c):
        c = c.flatten().tolist()
    if poses.is_cuda:
        from .. import ops
        return ops.gen_rays(poses, width, height, focal, z_near, z_far, c=c)
    from ..synthetic import gen_rays as _gr
    return _gr(poses, width, height, focal, z_near, z_far, c=c)
