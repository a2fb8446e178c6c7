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
    """src/util/util.py:279-323: camera-to-world pose on a sphere (degrees), Blender -> OpenGL axes."""
    def rot(axis, a):
        c, s_ = math.cos(a), math.sin(a)
        m = torch.eye(4)
        if axis == "phi":
            m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s_, s_, c
        else:
            m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, -s_, s_, c
        return m
    c2w = torch.eye(4)
    c2w[2, 3] = radius
    c2w = rot("phi", phi / 180.0 * math.pi) @ c2w
    c2w = rot("theta", theta / 180.0 * math.pi) @ c2w
    flip = torch.tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], dtype=torch.float32)
    return flip @ c2w


def unproj_map(width, height, f, c=None, device="cpu"):
    """src/util/util.py:113-143: unit camera-space ray per pixel (H,W,3), OpenGL convention (x right, y up, -z forward)."""
    if torch.is_tensor(f):
        f = float(f) if f.numel() == 1 else (float(f.flatten()[0]), float(f.flatten()[1]))
    if torch.is_tensor(c):
        c = c.flatten().tolist()
    if c is None:
        c = [width * 0.5, height * 0.5]
    fx, fy = (float(f), float(f)) if isinstance(f, (float, int)) else (float(f[0]), float(f[1]))
    ys = (torch.arange(height, dtype=torch.float32) - float(c[1])) / fy
    xs = (torch.arange(width, dtype=torch.float32) - float(c[0])) / fx
    d = torch.stack((xs[None, :].expand(height, -1), -ys[:, None].expand(-1, width), -torch.ones(height, width)), dim=-1)
    return (d / torch.norm(d, dim=-1, keepdim=True)).to(device)


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
    B = poses.shape[0]
    dirs = unproj_map(width, height, focal, c=c)[None].expand(B, -1, -1, -1)
    centers = poses[:, None, None, :3, 3].expand(-1, height, width, -1)
    raydir = torch.matmul(poses[:, None, None, :3, :3], dirs.unsqueeze(-1))[..., 0]
    nf = torch.tensor([float(z_near), float(z_far)]).expand(B, height, width, 2)
    return torch.cat((centers, raydir, nf), dim=-1)
