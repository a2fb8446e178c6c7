"""
torch.Tensor front-end of the C ABI (include/pixelnerf_hip.h).  PyTorch is plumbing here:
device memory, the current HIP stream and tensor shapes; every computation below is a call
into libpixelnerf_hip.so.  Nothing in this module has a CPU or eager-PyTorch fallback.
"""
import ctypes

import torch

from . import _lib

_MLP_KEYS = (
    ["lin_in.weight", "lin_in.bias", "lin_out.weight", "lin_out.bias"]
    + [f"lin_z.{b}.{s}" for b in range(3) for s in ("weight", "bias")]
    + [f"blocks.{b}.fc_{j}.{s}" for b in range(5) for j in (0, 1) for s in ("weight", "bias")]
)
_MLP_SHAPES = {"lin_in.weight": (512, 42), "lin_out.weight": (4, 512), "lin_out.bias": (4,)}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32(t, name, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise _lib.PixelNerfHipError(f"{name}: tensor must live on a HIP device (got {t.device}); "
                                     "pixelnerf_amd has no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if shape is not None:
        if len(shape) != t.dim() or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(t.shape)}")
    return t.contiguous()


class PackedMLP:
    """A ResnetFC's parameters repacked into the fused kernel's fragment stream.
    precision "f32" (exact validation path) keeps the raw nn.Linear tensors instead."""

    def __init__(self, buf, precision, weights=None, folded=False):
        self.buf = buf
        self.precision = precision
        self.weights = weights  # f32 only: (PnrMlpWeights, {key: tensor} keeping the storage alive)
        self.folded = folded    # stream without the lin_z GEMMs: must be used with fold_latent() tables

    @property
    def ptr(self):
        if self.buf is None:
            raise _lib.PixelNerfHipError("this entry point has no fp32 instantiation (precision='f32' covers "
                                         "inference through eval_ray_samples / eval_points / render_forward)")
        return ctypes.c_void_p(self.buf.data_ptr())

    @property
    def wref(self):
        return ctypes.byref(self.weights[0])


F32_CHUNK_POINTS = 1 << 17  # points per chunk of the fp32 path (workspace ~1.4 GB x NS)


def _f32_workspace(lib, scene, P):
    chunk = max(64, min(int(P), F32_CHUNK_POINTS // scene.NS))
    nbytes = lib.pnr_eval_f32_workspace_bytes(scene.NS, chunk)
    return torch.empty(nbytes, dtype=torch.uint8, device=scene.device), nbytes


def _weights_struct(state, combine_max=False):
    """-> (PnrMlpWeights of the tensors' device pointers, the tensors).  combine_max: the network pools its source views with
    util.combine_interleaved's "max" instead of the mean (src/util/util.py:461-471; ResnetFC.combine_type)."""
    keep = {}
    for k in _MLP_KEYS:
        if k not in state:
            raise KeyError(f"pack_mlp: missing parameter '{k}' (only the shipped ResnetFC shape "
                           "d_hidden=512, n_blocks=5, combine_layer=3 is supported)")
        shape = _MLP_SHAPES.get(k, (512, 512) if k.endswith("weight") else (512,))
        keep[k] = _f32(state[k].detach(), k, shape)
    w = _lib.PnrMlpWeights()
    w.lin_in_w, w.lin_in_b = keep["lin_in.weight"].data_ptr(), keep["lin_in.bias"].data_ptr()
    w.lin_out_w, w.lin_out_b = keep["lin_out.weight"].data_ptr(), keep["lin_out.bias"].data_ptr()
    for b in range(3):
        w.lin_z_w[b] = keep[f"lin_z.{b}.weight"].data_ptr()
        w.lin_z_b[b] = keep[f"lin_z.{b}.bias"].data_ptr()
    for b in range(5):
        w.fc0_w[b] = keep[f"blocks.{b}.fc_0.weight"].data_ptr()
        w.fc0_b[b] = keep[f"blocks.{b}.fc_0.bias"].data_ptr()
        w.fc1_w[b] = keep[f"blocks.{b}.fc_1.weight"].data_ptr()
        w.fc1_b[b] = keep[f"blocks.{b}.fc_1.bias"].data_ptr()
    w.combine_max = 1 if combine_max else 0
    return w, keep


def pack_mlp(state, precision="f16", backward=False, folded=False, weights=None, out=None, combine_max=False):
    """state: {reference ResnetFC state_dict key: float32 HIP tensor}
    (src/model/resnetfc.py:66-130: lin_in, lin_out, blocks.N.fc_0/fc_1, lin_z.N).
    backward=True packs the transposed streams of the data-gradient chain instead;
    folded=True packs the inference stream without the lin_z GEMMs (use with fold_latent).
    weights: a cached _weights_struct(state) result (the struct only holds pointers: it stays valid while the parameters
    are updated in place); out: a PackedMLP of the same form whose buffer is overwritten (same stream: ordered)."""
    lib = _lib.load()
    prec = _lib.PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    w, keep = weights if weights is not None else _weights_struct(state, combine_max)
    if prec == _lib.PREC_F32:
        if backward:
            raise _lib.PixelNerfHipError("precision='f32' has no backward path")
        return PackedMLP(None, prec, weights=(w, keep))
    dev = keep["lin_in.weight"].device
    if prec == _lib.PREC_F16X3:
        # fp32-class split-operand form: head + tail streams of the folded network (always used with fp32 tables)
        if backward:
            raise _lib.PixelNerfHipError("precision='f16x3' is an inference form (no backward streams)")
        buf = torch.empty(lib.pnr_packed_mlp_split_bytes(), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.pnr_pack_mlp_split(ctypes.byref(w), _p(buf), _stream()), "pnr_pack_mlp_split")
        return PackedMLP(buf, prec, folded=True)
    if backward and folded:
        raise ValueError("the backward streams have no folded form")
    nbytes = lib.pnr_packed_mlp_bwd_bytes() if backward else lib.pnr_packed_mlp_bytes()
    reuse = (out is not None and out.buf is not None and out.precision == prec and out.folded == bool(folded)
             and out.buf.numel() == nbytes and out.buf.device == dev)
    buf = out.buf if reuse else torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        if backward:
            _lib.check(lib.pnr_pack_mlp_bwd(ctypes.byref(w), prec, _p(buf), _stream()), "pnr_pack_mlp_bwd")
        elif folded:
            _lib.check(lib.pnr_pack_mlp_folded(ctypes.byref(w), prec, _p(buf), _stream()), "pnr_pack_mlp_folded")
        else:
            _lib.check(lib.pnr_pack_mlp(ctypes.byref(w), prec, _p(buf), _stream()), "pnr_pack_mlp")
    return out if reuse else PackedMLP(buf, prec, folded=folded)


def fold_latent(scene, state, precision="f16"):
    """Per-texel tables T_b = W_z[b] . grid + b_z[b] (b = 0..2) of one network for one encoded scene:
    (3, SB*NS, Hl, Wl, 512) 16-bit, hidden features in storage order.  Re-run when the grid or lin_z change."""
    lib = _lib.load()
    prec = _lib.PRECISIONS[precision] if isinstance(precision, str) else int(precision)
    if prec == _lib.PREC_F32:
        raise _lib.PixelNerfHipError("precision='f32' has no folded form")
    w, keep = _weights_struct(state)
    NV, Hl, Wl, _ = scene.latent_nhwc.shape
    if prec == _lib.PREC_F16X3:  # fp32 tables
        tables = torch.empty((3, NV, Hl, Wl, 512), dtype=torch.float32, device=scene.device)
        assert tables.numel() * 4 == lib.pnr_folded_tables_f32_bytes(scene.ref)
        with torch.cuda.device(scene.device):
            _lib.check(lib.pnr_fold_latent_f32(scene.ref, ctypes.byref(w), _p(tables), _stream()), "pnr_fold_latent_f32")
        return tables
    dt = torch.float16 if prec == _lib.PREC_F16 else torch.bfloat16
    tables = torch.empty((3, NV, Hl, Wl, 512), dtype=dt, device=scene.device)
    assert tables.numel() * 2 == lib.pnr_folded_tables_bytes(scene.ref)
    with torch.cuda.device(scene.device):
        _lib.check(lib.pnr_fold_latent(scene.ref, ctypes.byref(w), prec, _p(tables), _stream()), "pnr_fold_latent")
    return tables


def fold_latent_rows(scene, state, rays, z, tables):
    """The 'f16x3' tables of fold_latent() for the texels ONE training pass reads (pnr_fold_latent_f32_rows): rays (R,8), z (R,K) =
    the pass's samples.  Writes the marked rows of `tables` ((3, SB*NS, Hl, Wl, 512) fp32, zero-initialised once by the caller) in
    place; use the buffer only for an eval_ray_samples_split_train call on the same rays and z.  Grids of >= 8192 texels."""
    lib = _lib.load()
    w, keep = _weights_struct(state)
    NV, Hl, Wl, _ = scene.latent_nhwc.shape
    if tables.dtype != torch.float32 or tuple(tables.shape) != (3, NV, Hl, Wl, 512) or not tables.is_contiguous():
        raise _lib.PixelNerfHipError("fold_latent_rows: tables must be a contiguous (3, SB*NS, Hl, Wl, 512) fp32 tensor")
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    nbytes = int(lib.pnr_fold_latent_f32_rows_workspace_bytes(scene.ref))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=scene.device)
    with torch.cuda.device(scene.device):
        _lib.check(lib.pnr_fold_latent_f32_rows(scene.ref, ctypes.byref(w), _p(rays), _p(z), R, max(R // scene.SB, 1), z.shape[1], _p(tables),
                                                _p(ws), nbytes, _stream()), "pnr_fold_latent_f32_rows")
    return tables


def _check_fold(packed, tables, what):
    if packed.folded != (tables is not None):
        raise _lib.PixelNerfHipError(f"{what}: a folded network stream needs its fold_latent() tables (and only it takes them)")


def _check_split_tables(tables):
    if tables is None or tables.dtype != torch.float32:
        raise _lib.PixelNerfHipError("precision='f16x3' takes the fp32 tables of fold_latent(scene, state, 'f16x3')")


class Scene:
    """Device-side encoded-scene state (what PixelNeRFNet.encode() leaves behind)."""

    def __init__(self, latent_nhwc, poses, focal, c, image_shape, NS):
        NV, Hl, Wl, C = latent_nhwc.shape
        if C != 512:
            raise ValueError("latent must have 512 channels (encoder.latent_size of the shipped configs)")
        if NV % NS != 0:
            raise ValueError("latent rows must be SB*NS")
        self.latent_nhwc = _f32(latent_nhwc, "latent_nhwc")
        self.poses = _f32(poses, "poses", (NV, 3, 4))
        self.focal = _f32(focal, "focal", (None, 2))
        self.c = _f32(c, "c", (None, 2))
        self.NS, self.SB = int(NS), NV // int(NS)
        for nm, t in (("focal", self.focal), ("c", self.c)):
            if t.shape[0] not in (1, self.SB):
                raise ValueError(f"{nm} must have 1 or SB rows")
        s = _lib.PnrScene()
        s.latent_nhwc, s.poses = self.latent_nhwc.data_ptr(), self.poses.data_ptr()
        s.focal, s.c = self.focal.data_ptr(), self.c.data_ptr()
        s.SB, s.NS, s.Hl, s.Wl = self.SB, self.NS, Hl, Wl
        s.n_focal, s.n_c = self.focal.shape[0], self.c.shape[0]
        s.img_w, s.img_h = float(image_shape[0]), float(image_shape[1])
        self.struct = s
        self.device = self.latent_nhwc.device

    @property
    def ref(self):
        return ctypes.byref(self.struct)


def nchw_to_nhwc(latent):
    lib = _lib.load()
    latent = _f32(latent, "latent")
    N, C, H, W = latent.shape
    out = torch.empty((N, H, W, C), dtype=torch.float32, device=latent.device)
    with torch.cuda.device(latent.device):
        _lib.check(lib.pnr_nchw_to_nhwc(_p(latent), _p(out), N, C, H, W, _stream()), "pnr_nchw_to_nhwc")
    return out


def make_scene(latent_nchw, poses, focal, c, image_shape, NS):
    """latent_nchw (SB*NS,512,Hl,Wl) as stored in encoder.latent."""
    return Scene(nchw_to_nhwc(latent_nchw), poses, focal, c, image_shape, NS)


def sample_coarse(rays, u1, lindisp=False):
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    u1 = _f32(u1, "u1", (R, None))
    Kc = u1.shape[1]
    z = torch.empty((R, Kc), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_sample_coarse(_p(rays), _p(u1), R, Kc, int(lindisp), _p(z), _stream()),
                   "pnr_sample_coarse")
    return z


def sample_fine(rays, weights_c, depth_c, z_coarse, u2, u3, n4, depth_std=0.01, lindisp=False, want_ranks=False):
    """-> z_sorted (R, Kc + Kimp + Kfd).  u2/u3 may be None (no importance samples), n4 may be
    None (no depth samples).  want_ranks: also return (R,Kfd) int32 positions of the depth samples
    in z_sorted."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z_coarse = _f32(z_coarse, "z_coarse", (R, None))
    Kc = z_coarse.shape[1]
    Kimp = 0 if u2 is None else u2.shape[1]
    Kfd = 0 if n4 is None else n4.shape[1]
    if Kimp:
        weights_c = _f32(weights_c, "weights_c", (R, Kc))
        u2, u3 = _f32(u2, "u2", (R, Kimp)), _f32(u3, "u3", (R, Kimp))
    if Kfd:
        depth_c, n4 = _f32(depth_c, "depth_c", (R,)), _f32(n4, "n4", (R, Kfd))
    z = torch.empty((R, Kc + Kimp + Kfd), dtype=torch.float32, device=rays.device)
    ranks = torch.empty((R, Kfd), dtype=torch.int32, device=rays.device) if (want_ranks and Kfd) else None
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_sample_fine(_p(rays), _p(weights_c) if Kimp else None, _p(depth_c) if Kfd else None,
                                       _p(z_coarse), _p(u2) if Kimp else None, _p(u3) if Kimp else None,
                                       _p(n4) if Kfd else None, R, Kc, Kimp, Kfd, float(depth_std),
                                       int(lindisp), _p(z), _p(ranks), _stream()), "pnr_sample_fine")
    return (z, ranks) if want_ranks else z


def eval_ray_samples(scene, packed, rays, z, tables=None):
    """rays (R,8), z (R,K) -> rgbsigma (R,K,4); R = SB * rays_per_obj.  tables: fold_latent() output when
    `packed` is a folded stream."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    if R % scene.SB != 0:
        raise ValueError("number of rays must be a multiple of the number of objects")
    out = torch.empty((R, K, 4), dtype=torch.float32, device=rays.device)
    if packed.precision == _lib.PREC_F32:
        ws, nbytes = _f32_workspace(lib, scene, R * K)
        with torch.cuda.device(rays.device):
            _lib.check(lib.pnr_eval_ray_samples_f32(scene.ref, packed.wref, _p(rays), _p(z), R, max(R // scene.SB, 1), K,
                                                    _p(out), _p(ws), nbytes, _stream()), "pnr_eval_ray_samples_f32")
        return out
    _check_fold(packed, tables, "eval_ray_samples")
    if packed.precision == _lib.PREC_F16X3:
        _check_split_tables(tables)
        with torch.cuda.device(rays.device):
            _lib.check(lib.pnr_eval_ray_samples_split(scene.ref, packed.ptr, _p(tables), _p(rays), _p(z), R,
                                                      max(R // scene.SB, 1), K, _p(out), _stream()), "pnr_eval_ray_samples_split")
        return out
    with torch.cuda.device(rays.device):
        if tables is not None:
            _lib.check(lib.pnr_eval_ray_samples_folded(scene.ref, packed.ptr, _p(tables), packed.precision, _p(rays), _p(z), R,
                                                       max(R // scene.SB, 1), K, _p(out), _stream()),
                       "pnr_eval_ray_samples_folded")
        else:
            _lib.check(lib.pnr_eval_ray_samples(scene.ref, packed.ptr, packed.precision, _p(rays), _p(z), R,
                                                max(R // scene.SB, 1), K, _p(out), _stream()),
                       "pnr_eval_ray_samples")
    return out


RESNETFC_CHUNK_ROWS = 1 << 17  # rows per launch set of resnetfc_forward (workspace 0.5 GB)


def resnetfc_forward(state, zx, combine_inner_dims=(1,), combine_max=False):
    """ResnetFC.forward on explicit rows (src/model/resnetfc.py:132-184): zx (rows, 554) fp32 = [latent | code+viewdir],
    combine_inner_dims = (1,) or (NS, B) with rows ordered [group][view][point]; returns lin_out's raw output
    (rows / NS, 4).  Unfused fp32 linears (the exact-fp32 path's kernels); whole (NS, B) groups per launch set."""
    lib = _lib.load()
    zx = _f32(zx, "zx")
    if zx.dim() != 2 or zx.shape[1] != 554:
        raise ValueError("resnetfc_forward: zx must be (rows, 512 + 42)")
    rows = zx.shape[0]
    dims = tuple(int(d) for d in combine_inner_dims)
    if dims == (1,):
        NS, B = 1, 1
    elif len(dims) == 2:
        NS, B = dims
    else:
        raise ValueError("resnetfc_forward: combine_inner_dims must be (1,) or (NS, B)")
    if rows % (NS * B) != 0:
        raise ValueError("resnetfc_forward: rows must be a multiple of NS * B")
    w, keep = _weights_struct(state, combine_max)
    out = torch.empty((rows // NS, 4), dtype=torch.float32, device=zx.device)
    group = NS * B
    step = max(group, RESNETFC_CHUNK_ROWS // group * group)
    nbytes = lib.pnr_resnetfc_forward_f32_workspace_bytes(min(step, rows) if rows else group, NS)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=zx.device)
    with torch.cuda.device(zx.device):
        for r0 in range(0, rows, step):
            n = min(step, rows - r0)
            _lib.check(lib.pnr_resnetfc_forward_f32(ctypes.byref(w), _p(zx[r0:r0 + n]), n, NS, B, _p(out[r0 // NS:(r0 + n) // NS]),
                                                    _p(ws), nbytes, _stream()), "pnr_resnetfc_forward_f32")
    del keep
    return out


def eval_points(scene, packed, xyz, viewdirs, tables=None):
    """xyz, viewdirs (SB,B,3) -> (SB,B,4)."""
    lib = _lib.load()
    xyz = _f32(xyz, "xyz", (scene.SB, None, 3))
    B = xyz.shape[1]
    viewdirs = _f32(viewdirs, "viewdirs", (scene.SB, B, 3))
    out = torch.empty((scene.SB, B, 4), dtype=torch.float32, device=xyz.device)
    if packed.precision == _lib.PREC_F32:
        ws, nbytes = _f32_workspace(lib, scene, scene.SB * B)
        with torch.cuda.device(xyz.device):
            _lib.check(lib.pnr_eval_points_f32(scene.ref, packed.wref, _p(xyz), _p(viewdirs), B, _p(out), _p(ws), nbytes,
                                               _stream()), "pnr_eval_points_f32")
        return out
    _check_fold(packed, tables, "eval_points")
    if packed.precision == _lib.PREC_F16X3:
        _check_split_tables(tables)
        with torch.cuda.device(xyz.device):
            _lib.check(lib.pnr_eval_points_split(scene.ref, packed.ptr, _p(tables), _p(xyz), _p(viewdirs), B, _p(out), _stream()),
                       "pnr_eval_points_split")
        return out
    with torch.cuda.device(xyz.device):
        if tables is not None:
            _lib.check(lib.pnr_eval_points_folded(scene.ref, packed.ptr, _p(tables), packed.precision, _p(xyz), _p(viewdirs),
                                                  B, _p(out), _stream()), "pnr_eval_points_folded")
        else:
            _lib.check(lib.pnr_eval_points(scene.ref, packed.ptr, packed.precision, _p(xyz), _p(viewdirs), B,
                                           _p(out), _stream()), "pnr_eval_points")
    return out


def point_features(scene, xyz, viewdirs):
    """The fp32 feature phase alone: xyz, viewdirs (SB,B,3) -> (in42 (NS,SB,B,64) = [positional code(39) | rotated
    viewdir(3) | 0 pad], zlat (NS,SB,B,512) = SpatialEncoder.index of the projected points)."""
    lib = _lib.load()
    xyz = _f32(xyz, "xyz", (scene.SB, None, 3))
    B = xyz.shape[1]
    viewdirs = _f32(viewdirs, "viewdirs", (scene.SB, B, 3))
    in42 = torch.empty((scene.NS, scene.SB, B, 64), dtype=torch.float32, device=xyz.device)
    zlat = torch.empty((scene.NS, scene.SB, B, 512), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(lib.pnr_point_features_f32(scene.ref, _p(xyz), _p(viewdirs), B, _p(in42), _p(zlat), _stream()),
                   "pnr_point_features_f32")
    return in42, zlat


def composite(rays, z, rgbsigma, white_bkgd=False, want_weights=True):
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    rgbsigma = _f32(rgbsigma, "rgbsigma", (R, K, 4))
    dev = rays.device
    weights = torch.empty((R, K), dtype=torch.float32, device=dev) if want_weights else None
    rgb = torch.empty((R, 3), dtype=torch.float32, device=dev)
    depth = torch.empty((R,), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_composite(_p(rays), _p(z), _p(rgbsigma), R, K, int(bool(white_bkgd)), _p(weights),
                                     _p(rgb), _p(depth), _stream()), "pnr_composite")
    return weights, rgb, depth


def philox_noise(R, n_coarse, n_fine, n_fine_depth, seed, device, ray_id_offset=0, ray_id_stride=0, rays_per_obj=0):
    """The draws the seeded renderer entries make in-kernel, written out as the explicit noise dict
    (u1 (R,Kc)[, u2, u3 (R,Kf-Kfd)][, n4 (R,Kfd)]): Philox4x32-10 keyed by `seed`, counter (global ray id, index/4, draw)."""
    lib = _lib.load()
    Kc, Kf, Kfd = int(n_coarse), int(n_fine), int(n_fine_depth)
    Kimp = max(Kf - Kfd, 0) if Kf > 0 else 0
    Kfd = Kfd if Kf > 0 else 0
    out = {"u1": torch.empty((R, Kc), dtype=torch.float32, device=device)}
    if Kimp > 0:
        out["u2"] = torch.empty((R, Kimp), dtype=torch.float32, device=device)
        out["u3"] = torch.empty((R, Kimp), dtype=torch.float32, device=device)
    if Kfd > 0:
        out["n4"] = torch.empty((R, Kfd), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.pnr_philox_noise(int(seed) & (2 ** 64 - 1), int(ray_id_offset), int(ray_id_stride), int(rays_per_obj) or R,
                                        R, Kc, Kimp, Kfd, _p(out["u1"]), _p(out.get("u2")), _p(out.get("u3")), _p(out.get("n4")),
                                        _stream()), "pnr_philox_noise")
    return out


def _render_outputs(R, Kc, Kf, want_weights, dev):
    def outs(K):
        return (torch.empty((R, 3), dtype=torch.float32, device=dev),
                torch.empty((R,), dtype=torch.float32, device=dev),
                torch.empty((R, K), dtype=torch.float32, device=dev) if want_weights else None)
    return outs(Kc), (outs(Kc + Kf) if Kf > 0 else (None, None, None))


def _render_result(c, f, Kf, want_weights):
    ret = {"coarse": {"rgb": c[0], "depth": c[1]}}
    if want_weights:
        ret["coarse"]["weights"] = c[2]
    if Kf > 0:
        ret["fine"] = {"rgb": f[0], "depth": f[1]}
        if want_weights:
            ret["fine"]["weights"] = f[2]
    return ret


def _explicit_noise(noise, R, Kc, Kf, Kfd):
    Kimp = Kf - Kfd
    u1 = _f32(noise["u1"], "u1", (R, Kc))
    u2 = u3 = n4 = None
    if Kf > 0 and Kimp > 0:
        u2, u3 = _f32(noise["u2"], "u2", (R, Kimp)), _f32(noise["u3"], "u3", (R, Kimp))
    if Kf > 0 and Kfd > 0:
        n4 = _f32(noise["n4"], "n4", (R, Kfd))
    return u1, u2, u3, n4


def render_forward(scene, packed_coarse, packed_fine, rays, n_coarse, n_fine, n_fine_depth, noise,
                   depth_std=0.01, white_bkgd=False, lindisp=False, want_weights=False, tables=None,
                   seed=0, ray_id_offset=0, ray_id_stride=0):
    """Whole NeRFRenderer.forward (nerf.py:251-303) for rays (R,8) in object-major order.
    noise: dict(u1[,u2,u3][,n4]) of pre-drawn tensors, or None: the sampling kernels draw from the counter-based
    generator keyed by `seed` (ops.philox_noise gives the same values as tensors); ray_id_offset / ray_id_stride place a
    shard of rays inside the whole ray set so that a sharded render equals the unsharded one.
    Returns {"coarse": {...}, "fine": {...}} of flat tensors.
    tables: (tables_coarse, tables_fine|None) from fold_latent() when the networks are folded streams."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R, dev = rays.shape[0], rays.device
    Kc, Kf, Kfd = int(n_coarse), int(n_fine), int(n_fine_depth)
    Kimp = Kf - Kfd
    if Kf > 0 and Kimp < 0:
        raise ValueError("n_fine_depth must not exceed n_fine")
    if R % scene.SB != 0:
        raise ValueError("number of rays must be a multiple of the number of objects")
    if packed_fine is not None and packed_fine.precision != packed_coarse.precision:
        raise ValueError("coarse and fine networks must be packed at the same precision")
    per_obj = max(R // scene.SB, 1)
    if packed_coarse.precision == _lib.PREC_F32:
        # exact-fp32 validation path: the same stages, one C call each
        if noise is None:
            noise = philox_noise(R, Kc, Kf, Kfd, seed, dev, ray_id_offset, ray_id_stride, per_obj)
        u1, u2, u3, n4 = _explicit_noise(noise, R, Kc, Kf, Kfd)
        z_c = sample_coarse(rays, u1, lindisp)
        w_c, rgb_c, depth_c = composite(rays, z_c, eval_ray_samples(scene, packed_coarse, rays, z_c), white_bkgd, True)
        ret = {"coarse": {"rgb": rgb_c, "depth": depth_c}}
        if want_weights:
            ret["coarse"]["weights"] = w_c
        if Kf > 0:
            z_f = sample_fine(rays, w_c, depth_c, z_c, u2, u3, n4, depth_std, lindisp)
            fine = packed_fine if packed_fine is not None else packed_coarse
            w_f, rgb_f, depth_f = composite(rays, z_f, eval_ray_samples(scene, fine, rays, z_f), white_bkgd, want_weights)
            ret["fine"] = {"rgb": rgb_f, "depth": depth_f}
            if want_weights:
                ret["fine"]["weights"] = w_f
        return ret

    c, f = _render_outputs(R, Kc, Kf, want_weights, dev)
    ws = torch.empty(max(lib.pnr_render_workspace_bytes(R, Kc, Kf), 16), dtype=torch.uint8, device=dev)
    tc, tf = tables if tables is not None else (None, None)
    if packed_coarse.precision == _lib.PREC_F16X3:
        _check_split_tables(tc)
        if packed_fine is not None:
            _check_split_tables(tf)
    _check_fold(packed_coarse, tc, "render_forward")
    if packed_fine is not None:
        _check_fold(packed_fine, tf, "render_forward")
    pf = packed_fine.ptr if packed_fine is not None else None
    with torch.cuda.device(dev):
        if noise is None:
            _lib.check(lib.pnr_render_forward_seeded(
                scene.ref, packed_coarse.ptr, _p(tc), pf, _p(tf), packed_coarse.precision, _p(rays), R, per_obj, Kc, Kf, Kfd,
                float(depth_std), int(bool(white_bkgd)), int(bool(lindisp)), int(seed) & (2 ** 64 - 1), int(ray_id_offset),
                int(ray_id_stride), _p(c[0]), _p(c[1]), _p(c[2]), _p(f[0]), _p(f[1]), _p(f[2]), _p(ws), _stream()),
                "pnr_render_forward_seeded")
        else:
            u1, u2, u3, n4 = _explicit_noise(noise, R, Kc, Kf, Kfd)
            if tc is not None:
                _lib.check(lib.pnr_render_forward_folded(
                    scene.ref, packed_coarse.ptr, _p(tc), pf, _p(tf), packed_coarse.precision, _p(rays), R, per_obj, Kc, Kf, Kfd,
                    float(depth_std), int(bool(white_bkgd)), int(bool(lindisp)), _p(u1), _p(u2), _p(u3), _p(n4),
                    _p(c[0]), _p(c[1]), _p(c[2]), _p(f[0]), _p(f[1]), _p(f[2]), _p(ws), _stream()), "pnr_render_forward_folded")
            else:
                _lib.check(lib.pnr_render_forward(
                    scene.ref, packed_coarse.ptr, pf, packed_coarse.precision, _p(rays), R, per_obj, Kc, Kf, Kfd,
                    float(depth_std), int(bool(white_bkgd)), int(bool(lindisp)), _p(u1), _p(u2), _p(u3), _p(n4),
                    _p(c[0]), _p(c[1]), _p(c[2]), _p(f[0]), _p(f[1]), _p(f[2]), _p(ws), _stream()), "pnr_render_forward")
    return _render_result(c, f, Kf, want_weights)


def gen_rays(poses, width, height, focal, z_near, z_far, c=None):
    """util.gen_rays (src/util/util.py:238-276, ndc=False): poses (NV,4,4) -> (NV,H,W,8)."""
    lib = _lib.load()
    poses = _f32(poses, "poses", (None, 4, 4))
    NV = poses.shape[0]
    fx, fy = (float(focal), float(focal)) if not hasattr(focal, "__len__") else (float(focal[0]), float(focal[-1]))
    cx, cy = (width * 0.5, height * 0.5) if c is None else (float(c[0]), float(c[1]))
    rays = torch.empty((NV, height, width, 8), dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        _lib.check(lib.pnr_gen_rays(_p(poses), NV, int(width), int(height), fx, fy, cx, cy, float(z_near),
                                    float(z_far), _p(rays), _stream()), "pnr_gen_rays")
    return rays


def render_views(scene, packed_coarse, packed_fine, poses_c2w, width, height, focal, z_near, z_far, n_coarse, n_fine,
                 n_fine_depth, noise, c=None, depth_std=0.01, white_bkgd=False, lindisp=False, want_weights=False, tables=None,
                 seed=0):
    """util.gen_rays + NeRFRenderer.forward in one C call (eval/eval.py:247-279): poses_c2w (NV,4,4),
    views grouped per object; every pixel of every view is rendered, its ray regenerated inside the kernels (no ray
    array).  noise: explicit dict laid out for R = NV*H*W rays, or None for in-kernel draws from `seed`.
    tables: (tables_coarse, tables_fine|None) for folded streams.  Returns the same nested dict as render_forward with
    R = NV*H*W rows (reshape to (NV,H,W,...))."""
    lib = _lib.load()
    poses = _f32(poses_c2w, "poses_c2w", (None, 4, 4))
    NV, W, H = poses.shape[0], int(width), int(height)
    R, dev = NV * W * H, poses.device
    if packed_coarse.precision == _lib.PREC_F32:
        rays = gen_rays(poses, W, H, focal, z_near, z_far, c).reshape(-1, 8)
        return render_forward(scene, packed_coarse, packed_fine, rays, n_coarse, n_fine, n_fine_depth, noise, depth_std,
                              white_bkgd, lindisp, want_weights, seed=seed)
    Kc, Kf, Kfd = int(n_coarse), int(n_fine), int(n_fine_depth)
    if Kf > 0 and Kf - Kfd < 0:
        raise ValueError("n_fine_depth must not exceed n_fine")
    if NV % scene.SB != 0:
        raise ValueError("number of views must be a multiple of the number of objects")
    fx, fy = (float(focal), float(focal)) if not hasattr(focal, "__len__") else (float(focal[0]), float(focal[-1]))
    cx, cy = (W * 0.5, H * 0.5) if c is None else (float(c[0]), float(c[1]))
    if packed_fine is not None and packed_fine.precision != packed_coarse.precision:
        raise ValueError("coarse and fine networks must be packed at the same precision")
    tc, tf = tables if tables is not None else (None, None)
    _check_fold(packed_coarse, tc, "render_views")  # a folded stream without its tables would read as garbage
    if packed_fine is not None:
        _check_fold(packed_fine, tf, "render_views")
    if packed_coarse.precision == _lib.PREC_F16X3:
        _check_split_tables(tc)
    u1 = u2 = u3 = n4 = None
    if noise is not None:
        u1, u2, u3, n4 = _explicit_noise(noise, R, Kc, Kf, Kfd)
    co, fo = _render_outputs(R, Kc, Kf, want_weights, dev)
    ws = torch.empty(max(lib.pnr_render_views_workspace_bytes(NV, W, H, Kc, Kf), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_render_views(
            scene.ref, packed_coarse.ptr, _p(tc), packed_fine.ptr if packed_fine is not None else None, _p(tf),
            packed_coarse.precision, _p(poses), NV, W, H, fx, fy, cx, cy, float(z_near), float(z_far), Kc, Kf, Kfd,
            float(depth_std), int(bool(white_bkgd)), int(bool(lindisp)), _p(u1), _p(u2), _p(u3), _p(n4), int(seed) & (2 ** 64 - 1),
            _p(co[0]), _p(co[1]), _p(co[2]), _p(fo[0]), _p(fo[1]), _p(fo[2]), _p(ws), _stream()), "pnr_render_views")
    return _render_result(co, fo, Kf, want_weights)


def positional_encoding(x, freqs2, phases2, include_input=True):
    """PositionalEncoding.forward (src/model/code.py:30-42) through pnr_positional_encoding.  x (N, d_in) float32 HIP tensor;
    freqs2 / phases2: the module's `_freqs` / `_phases` buffers (2 * num_freqs values each).  -> (N, d_out)."""
    lib = _lib.load()
    x = _f32(x, "x", (None, None))
    N, d_in = x.shape
    freqs2 = _f32(freqs2.reshape(-1), "freqs2", (None,))
    phases2 = _f32(phases2.reshape(-1), "phases2", (freqs2.shape[0],))
    if freqs2.shape[0] % 2:
        raise ValueError("freqs2 holds every frequency twice (sin and cos as a phase-shifted sin)")
    F = freqs2.shape[0] // 2
    out = torch.empty((N, d_in * (2 * F + (1 if include_input else 0))), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.pnr_positional_encoding(_p(x), N, d_in, F, _p(freqs2), _p(phases2), int(bool(include_input)), _p(out),
                                               _stream()), "pnr_positional_encoding")
    return out


def positional_encoding_backward(x, g_out, freqs2, phases2, include_input=True):
    """Gradient of positional_encoding with respect to x (pnr_positional_encoding_backward).  -> (N, d_in)."""
    lib = _lib.load()
    x = _f32(x, "x", (None, None))
    N, d_in = x.shape
    freqs2 = _f32(freqs2.reshape(-1), "freqs2", (None,))
    phases2 = _f32(phases2.reshape(-1), "phases2", (freqs2.shape[0],))
    F = freqs2.shape[0] // 2
    g_out = _f32(g_out, "g_out", (N, d_in * (2 * F + (1 if include_input else 0))))
    gx = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(lib.pnr_positional_encoding_backward(_p(x), _p(g_out), N, d_in, F, _p(freqs2), _p(phases2),
                                                        int(bool(include_input)), _p(gx), _stream()),
                   "pnr_positional_encoding_backward")
    return gx


def grid_index(latent_nhwc, uv):
    """SpatialEncoder.index's lookup (src/model/encoder.py:100-109) through pnr_grid_index: latent_nhwc (NV,Hl,Wl,C) channel-last
    grid, uv (NV,N,2) NORMALISED coordinates in [-1,1] -> (NV,C,N) as F.grid_sample(bilinear, border, align_corners=True)."""
    lib = _lib.load()
    latent_nhwc = _f32(latent_nhwc, "latent_nhwc", (None, None, None, None))
    NV, Hl, Wl, C = latent_nhwc.shape
    uv = _f32(uv, "uv", (NV, None, 2))
    N = uv.shape[1]
    out = torch.empty((NV, C, N), dtype=torch.float32, device=uv.device)
    with torch.cuda.device(uv.device):
        _lib.check(lib.pnr_grid_index(_p(latent_nhwc), NV, Hl, Wl, C, _p(uv), N, _p(out), _stream()), "pnr_grid_index")
    return out


def grid_index_backward(latent_nhwc, uv, g_out, want_latent=True, want_uv=True):
    """-> (d_latent_nhwc (NV,Hl,Wl,C) | None, d_uv (NV,N,2) | None) of grid_index (pnr_grid_index_backward)."""
    lib = _lib.load()
    latent_nhwc = _f32(latent_nhwc, "latent_nhwc", (None, None, None, None))
    NV, Hl, Wl, C = latent_nhwc.shape
    uv = _f32(uv, "uv", (NV, None, 2))
    N = uv.shape[1]
    g_out = _f32(g_out, "g_out", (NV, C, N))
    d_lat = torch.zeros_like(latent_nhwc) if want_latent else None
    d_uv = torch.empty_like(uv) if want_uv else None
    with torch.cuda.device(uv.device):
        _lib.check(lib.pnr_grid_index_backward(_p(latent_nhwc), NV, Hl, Wl, C, _p(uv), N, _p(g_out), _p(d_lat), _p(d_uv), _stream()),
                   "pnr_grid_index_backward")
    return d_lat, d_uv


# ---------------------------------------------------------------- fp16-range guard of the fp32-class kernels
# include/pixelnerf_hip.h, pnr_saturation_guard: while armed, the split-operand network kernels note every layer whose operand
# image received a value >= 65504 (fp16 heads saturate there: the result leaves the reference's fp32 arithmetic class) or whose
# output is not finite.  The flag words travel to pinned host memory with an asynchronous copy and are looked at on a later
# call -- no host synchronisation, like the parameter content check of ResnetFC.
_SAT = {}
SAT_LAYER_NAMES = ([f"relu(x) entering blocks.{b}.fc_0" if i == 0 else f"relu(net) entering blocks.{b}.fc_1" for b in range(5) for i in (0, 1)]
                   + ["the stream in front of lin_out", "a non-finite network output", "the feature grid / lin_z weights of the per-texel fold"])


def _sat_state(device, owner=None):
    """flag words + pinned host copies, per (device, owner): two networks on one device do not read each other's verdicts.
    pending: [(event, pinned words)] of guarded calls whose flag copy is still on its way; carry: bits of copies that arrived
    but were not handed to a poll yet.  A guarded call never overwrites the words of an earlier one."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, owner)
    st = _SAT.get(key)
    if st is None:
        st = dict(flags=torch.zeros(2, dtype=torch.int32, device=torch.device("cuda", idx)),
                  pool=[], pending=[], carry=[0, 0], seen=False, armed=False)
        _SAT[key] = st
    return st


def _sat_harvest(st, wait=False):
    """fold every ARRIVED flag copy into st["carry"] (wait=True: all of them), hand its pinned words back to the pool"""
    still = []
    for ev, host in st["pending"]:
        if wait:
            ev.synchronize()
        if wait or ev.query():
            w = host.tolist()
            st["carry"][0] |= int(w[0]) & 0xFFFFFFFF
            st["carry"][1] |= int(w[1]) & 0xFFFFFFFF
            st["seen"] = True
            st["pool"].append(host)
        else:
            still.append((ev, host))
    st["pending"] = still


def saturation_guard_release(owner):
    """drop the guard state of an owner that goes away (PixelNeRFNet.__del__)"""
    for key in [k for k in _SAT if k[1] == owner]:
        _SAT.pop(key, None)


def saturation_guard_arm(device, owner=None):
    """arm the guard for the launches this host thread makes next on `device` (until saturation_guard_disarm)"""
    st = _sat_state(device, owner)
    _lib.check(_lib.load().pnr_saturation_guard(_p(st["flags"])), "pnr_saturation_guard")
    st["armed"] = True


def saturation_guard_slot(device, slot, owner=None):
    """while armed: direct network launches (pnr_eval_*_split*) report into word `slot` (0 = coarse network, 1 = fine network);
    the render entries pick the word themselves"""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = next((v for (d, o), v in _SAT.items() if d == idx and v["armed"]), None) if owner is None else _sat_state(device, owner)
    if st is not None and st["armed"]:
        _lib.check(_lib.load().pnr_saturation_guard(ctypes.c_void_p(st["flags"].data_ptr() + 4 * (1 if slot else 0))), "pnr_saturation_guard")


def saturation_guard_disarm(device, owner=None):
    """disarm, and send the flag words on their way to the host (asynchronous; saturation_guard_poll reads them).  Every guarded
    call gets its own pinned words: under PIXELNERF_SATURATION_GUARD=always the previous call's copy is usually still in flight
    when the next one ends, and re-using one buffer lost its bits."""
    st = _sat_state(device, owner)
    _lib.check(_lib.load().pnr_saturation_guard(None), "pnr_saturation_guard")
    if st["armed"]:
        st["armed"] = False
        if len(st["pending"]) >= 8:  # bounded: fold the oldest copies in (blocks only when the GPU is 8 guarded calls behind)
            _sat_harvest(st, wait=True)
        host = st["pool"].pop() if st["pool"] else torch.zeros(2, dtype=torch.int32).pin_memory()
        with torch.cuda.device(st["flags"].device):
            host.copy_(st["flags"], non_blocking=True)
            st["flags"].zero_()
            ev = torch.cuda.Event()
            ev.record()
        st["pending"].append((ev, host))


def saturation_guard_poll(device, wait=False, owner=None):
    """-> (bits of the coarse-network launches, bits of the fine-network launches), OR-ed over the guarded calls whose flag copy
    has arrived since the last poll, or None when none has (wait=True blocks for every copy in flight).  Must not be called
    while the current stream is capturing (Event.query is illegal there): callers check that first."""
    st = _sat_state(device, owner)
    _sat_harvest(st, wait=wait)
    if not st["seen"]:
        return None
    bits = (st["carry"][0], st["carry"][1])
    st["carry"], st["seen"] = [0, 0], False
    return bits


def describe_saturation(bits):
    return ", ".join(SAT_LAYER_NAMES[i] for i in range(len(SAT_LAYER_NAMES)) if bits >> i & 1)


def pyramid_to_latent(stages, want_nchw=True):
    """Encoder output formatting (src/model/encoder.py:150-163): stages = list of (NV,C_s,H_s,W_s) float32
    HIP tensors (ResNet stage outputs).  -> (latent_nhwc (NV,H0,W0,sum C), latent_nchw (NV,sum C,H0,W0) | None):
    bilinear(align_corners=True) upsample to stage 0's size + channel concat, in one pass."""
    lib = _lib.load()
    stages = [_f32(t, f"stages[{i}]", (None, None, None, None)) for i, t in enumerate(stages)]
    n = len(stages)
    NV = stages[0].shape[0]
    if any(t.shape[0] != NV for t in stages):
        raise ValueError("all stages must have the same batch size")
    H0, W0 = stages[0].shape[2:]
    Ctot = sum(t.shape[1] for t in stages)
    dev = stages[0].device
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in stages])
    ch = (ctypes.c_int * n)(*[t.shape[1] for t in stages])
    hs = (ctypes.c_int * n)(*[t.shape[2] for t in stages])
    wd = (ctypes.c_int * n)(*[t.shape[3] for t in stages])
    nhwc = torch.empty((NV, H0, W0, Ctot), dtype=torch.float32, device=dev)
    nchw = torch.empty((NV, Ctot, H0, W0), dtype=torch.float32, device=dev) if want_nchw else None
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_pyramid_to_latent(ptrs, ch, hs, wd, n, NV, _p(nhwc), _p(nchw), _stream()),
                   "pnr_pyramid_to_latent")
    return nhwc, nchw


def sample_training_rays(poses, images, focal, z_near, z_far, ids, c=None, bboxes=None, ux=None, uy=None):
    """train/train.py:143-182 on device.  poses (SB,NV,4,4), images (SB,NV,3,H,W) in [-1,1], focal (SB,2),
    c (SB,2)|None; ids (SB,B) int64: view ids (with bboxes (SB,NV,4) + ux, uy (SB,B)) or flat pixel indices.
    -> rays (SB,B,8), rgb_gt (SB,B,3)."""
    lib = _lib.load()
    poses = _f32(poses, "poses", (None, None, 4, 4))
    SB, NV = poses.shape[:2]
    images = _f32(images, "images", (SB, NV, 3, None, None))
    H, W = images.shape[-2:]
    focal = _f32(focal, "focal", (SB, 2))
    c = None if c is None else _f32(c, "c", (SB, 2))
    if ids.dtype != torch.int64 or not ids.is_cuda or ids.dim() != 2 or ids.shape[0] != SB:
        raise TypeError("ids: expected an int64 HIP tensor of shape (SB,B)")
    ids = ids.contiguous()
    B = ids.shape[1]
    if bboxes is not None:
        bboxes = _f32(bboxes, "bboxes", (SB, NV, 4))
        ux, uy = _f32(ux, "ux", (SB, B)), _f32(uy, "uy", (SB, B))
    dev = poses.device
    rays = torch.empty((SB, B, 8), dtype=torch.float32, device=dev)
    gt = torch.empty((SB, B, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_sample_training_rays(_p(poses), _p(images), _p(focal), _p(c), _p(bboxes), _p(ids), _p(ux), _p(uy),
                                                SB, NV, W, H, B, float(z_near), float(z_far), _p(rays), _p(gt), _stream()),
                   "pnr_sample_training_rays")
    return rays, gt


def eval_epilogue(rgb, depth=None, z_near=0.0, z_far=1.0, gt_rgb=None, want_u8=True):
    """eval/eval.py:283-290,327-329 on device.  rgb (NV,P,3) [+ depth (NV,P)] [+ gt_rgb (NV,P,3) in [0,1]] ->
    dict(rgb (clamped), rgb_u8, depth_norm, sse (NV,) float64, psnr (NV,) float64)."""
    lib = _lib.load()
    rgb = _f32(rgb, "rgb", (None, None, 3))
    NV, P = rgb.shape[:2]
    dev = rgb.device
    depth = None if depth is None else _f32(depth, "depth", (NV, P))
    gt = None if gt_rgb is None else _f32(gt_rgb, "gt_rgb", (NV, P, 3))
    out = {"rgb": torch.empty_like(rgb)}
    u8 = torch.empty((NV, P, 3), dtype=torch.uint8, device=dev) if want_u8 else None
    dn = torch.empty((NV, P), dtype=torch.float32, device=dev) if depth is not None else None
    sse = torch.empty((NV,), dtype=torch.float64, device=dev) if gt is not None else None
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_eval_epilogue(_p(rgb), _p(depth), NV, P, float(z_near), float(z_far), _p(gt), _p(u8),
                                         _p(out["rgb"]), _p(dn), _p(sse), _stream()), "pnr_eval_epilogue")
    if want_u8:
        out["rgb_u8"] = u8
    if dn is not None:
        out["depth_norm"] = dn
    if sse is not None:
        out["sse"] = sse
        out["psnr"] = -10.0 * torch.log10(sse / (3 * P))
    return out


def profile_enable(on=True):
    _lib.check(_lib.load().pnr_profile_enable(int(bool(on))), "pnr_profile_enable")


def profile_read():
    ms, n = ctypes.c_double(0), ctypes.c_int(0)
    _lib.check(_lib.load().pnr_profile_read(ctypes.byref(ms), ctypes.byref(n)), "pnr_profile_read")
    return ms.value, n.value


def debug_set_x_dump(t):
    """test hook: dump the residual stream before lin_out of subsequent launches into t (P,512)."""
    _lib.check(_lib.load().pnr_debug_set_x_dump(_p(t)), "pnr_debug_set_x_dump")


PHASES = ["sync_top", "geometry", "gather", "gemm_in_z0", "bar1", "write_x", "bar2", "gemm_fc0", "bar3", "write_net",
          "bar4", "gemm_fc1_z", "lin_out", "bar_out", "final", "table", "own_bias", "own_prologue", "own_ksteps"]


def debug_phase_timing(scene, packed, rays, z, tables=None):
    """diagnostic: {phase: [s_memtime ticks of wave 0..7]} of workgroup 0, one f16 single-view launch."""
    lib = _lib.load()
    rays, z = _f32(rays, "rays", (None, 8)), _f32(z, "z")
    tim = torch.zeros(8 * len(PHASES), dtype=torch.int64, device=rays.device)
    _check_fold(packed, tables, "debug_phase_timing")
    _lib.check(lib.pnr_debug_phase_timing(scene.ref, packed.ptr, _p(tables), _p(rays), _p(z), rays.shape[0],
                                          max(rays.shape[0] // scene.SB, 1), z.shape[1], _p(tim), _stream()),
               "pnr_debug_phase_timing")
    torch.cuda.synchronize()
    t = tim.cpu().reshape(8, len(PHASES))
    return {p: t[:, i].tolist() for i, p in enumerate(PHASES)}  # per wave


def debug_phase_timing_split(scene, packed, rays, z, tables):
    """diagnostic: {phase: [cycles of wave 0..7]} of workgroup 0, one f16x3 single-view launch."""
    lib = _lib.load()
    rays, z = _f32(rays, "rays", (None, 8)), _f32(z, "z")
    _check_split_tables(tables)
    tim = torch.zeros(8 * len(PHASES), dtype=torch.int64, device=rays.device)
    out = torch.empty((rays.shape[0], z.shape[1], 4), dtype=torch.float32, device=rays.device)
    _lib.check(lib.pnr_debug_phase_timing_split(scene.ref, packed.ptr, _p(tables), _p(rays), _p(z), rays.shape[0],
                                                max(rays.shape[0] // scene.SB, 1), z.shape[1], _p(out), _p(tim), _stream()),
               "pnr_debug_phase_timing_split")
    torch.cuda.synchronize()
    t = tim.cpu().reshape(8, len(PHASES))
    return {p: t[:, i].tolist() for i, p in enumerate(PHASES)}


# ------------------------------------------------------------------ training support


def storage_perm(device=None):
    """LongTensor perm (512): perm[e] = hidden feature stored at position e of a dump row."""
    arr = (ctypes.c_int32 * 512)()
    _lib.check(_lib.load().pnr_storage_perm(arr), "pnr_storage_perm")
    return torch.tensor(list(arr), dtype=torch.long, device=device)


# Dump sets are recycled: a training step allocates the same ~50 tensors (1.2 GB at config-5 sizes) every step, and the
# host time of those allocations is comparable to the GPU time of a small kernel each.  acquire() hands out a free set of
# the right shape or builds one; the autograd Functions release() a set once its last consumer is enqueued (same-stream
# ordering makes the reuse safe).  A set that is never released is simply garbage-collected.
_DUMP_POOL = {}  # key -> free sets; insertion order = least recently released first
_DUMP_POOL_BYTES = [0]
DUMP_POOL_MAX_BYTES = 16 << 30  # sets beyond this total are not kept (they go back to torch's allocator)
DUMP_POOL_MAX_PER_KEY = 2       # a step holds at most the coarse and the fine set of a shape at once
DUMP_POOL_MAX_KEYS = 8          # shapes (ray-batch sizes) remembered: older ones are dropped -> a changing batch size cannot pile up sets


def _pool_get(key):
    free = _DUMP_POOL.get(key)
    if not free:
        return None
    obj = free.pop()
    _DUMP_POOL_BYTES[0] -= obj.nbytes
    return obj


def _pool_put(key, obj):
    free = _DUMP_POOL.pop(key, [])  # re-inserted below: most recently used last
    if len(free) < DUMP_POOL_MAX_PER_KEY and _DUMP_POOL_BYTES[0] + obj.nbytes <= DUMP_POOL_MAX_BYTES:
        free.append(obj)
        _DUMP_POOL_BYTES[0] += obj.nbytes
    if free:
        _DUMP_POOL[key] = free
    while len(_DUMP_POOL) > DUMP_POOL_MAX_KEYS:  # forget the least recently used shape
        old = next(iter(_DUMP_POOL))
        _DUMP_POOL_BYTES[0] -= sum(o.nbytes for o in _DUMP_POOL.pop(old))


def dump_pool_clear():
    """Drop the recycled dump sets (e.g. after the last training step of a process that goes on to do something else)."""
    _DUMP_POOL.clear()
    _DUMP_POOL_BYTES[0] = 0


class TrainDumps:
    """16-bit dumps of every linear layer's input operand for P points x NS views, plus their relu bit masks."""

    @classmethod
    def acquire(cls, P, NS, precision, device):
        key = ("fwd", int(P), int(NS), int(precision), str(device))
        obj = _pool_get(key)
        if obj is None:
            obj = cls(P, NS, precision, device)
            obj._pool_key = key
        obj._free = False
        return obj

    def release(self):
        if getattr(self, "_pool_key", None) is not None and not self._free:  # a second release of the same set is a no-op
            self._free = True
            _pool_put(self._pool_key, self)

    def __init__(self, P, NS, precision, device):
        dt = torch.float16 if precision == _lib.PREC_F16 else torch.bfloat16
        rv, rp = NS * P, P
        self.P, self.NS, self.dtype = P, NS, dt
        self.d_in = torch.empty((rv, 64), dtype=dt, device=device)
        self.d_z = torch.empty((rv, 512), dtype=dt, device=device)
        self.d_a = [torch.empty((rv if b < 3 else rp, 512), dtype=dt, device=device) for b in range(5)]
        self.d_n = [torch.empty((rv if b < 3 else rp, 512), dtype=dt, device=device) for b in range(5)]
        self.d_x5 = torch.empty((rp, 512), dtype=dt, device=device)
        self.d_mask = torch.empty(_lib.load().pnr_train_masks_bytes(P, NS), dtype=torch.uint8, device=device)
        s = _lib.PnrTrainDumps()
        s.d_in, s.d_z, s.d_x5 = self.d_in.data_ptr(), self.d_z.data_ptr(), self.d_x5.data_ptr()
        s.d_mask = self.d_mask.data_ptr()
        self.nbytes = sum(t.numel() * t.element_size() for t in [self.d_in, self.d_z, self.d_x5, self.d_mask] + self.d_a + self.d_n)
        for b in range(5):
            s.d_a[b], s.d_n[b] = self.d_a[b].data_ptr(), self.d_n[b].data_ptr()
        self.struct = s


class BackwardDumps:
    """Outputs of the fused data-gradient chain: every layer's dY (16-bit rows, operands of the weight-gradient GEMMs)
    plus, in fp32, d(interpolated latent) = sum_b dY_b W_z[b] and d(code | viewdir) = dY W_in."""

    @classmethod
    def acquire(cls, fwd, device):
        key = ("bwd", int(fwd.P), int(fwd.NS), str(fwd.dtype), str(device))
        obj = _pool_get(key)
        if obj is None:
            obj = cls(fwd, device)
            obj._pool_key = key
        obj._free = False
        return obj

    def release(self):
        if getattr(self, "_pool_key", None) is not None and not self._free:
            self._free = True
            _pool_put(self._pool_key, self)

    def __init__(self, fwd, device):
        self.g_fc1 = [torch.empty_like(t) for t in fwd.d_n]
        self.g_fc0 = [torch.empty_like(t) for t in fwd.d_a]
        self.g_x0 = torch.empty_like(fwd.d_z)
        rows = fwd.d_z.shape[0]
        self.d_zlat = torch.empty((rows, 512), dtype=torch.float32, device=device)
        self.d_in = torch.empty((rows, 42), dtype=torch.float32, device=device)
        self.nbytes = sum(t.numel() * t.element_size() for t in [self.g_x0, self.d_zlat, self.d_in] + self.g_fc1 + self.g_fc0)
        s = _lib.PnrBackwardDumps()
        s.g_x0 = self.g_x0.data_ptr()
        s.d_zlat, s.d_in = self.d_zlat.data_ptr(), self.d_in.data_ptr()
        for b in range(5):
            s.g_fc1[b], s.g_fc0[b] = self.g_fc1[b].data_ptr(), self.g_fc0[b].data_ptr()
        self.struct = s


def eval_ray_samples_train(scene, packed, rays, z):
    """eval_ray_samples + TrainDumps."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    dumps = TrainDumps.acquire(R * K, scene.NS, packed.precision, rays.device)
    out = torch.empty((R, K, 4), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_eval_ray_samples_train(scene.ref, packed.ptr, packed.precision, _p(rays), _p(z), R,
                                                  max(R // scene.SB, 1), K, _p(out), ctypes.byref(dumps.struct),
                                                  _stream()), "pnr_eval_ray_samples_train")
    return out, dumps


class F32Saved:
    """fp32 activations kept by the exact-fp32 training forward (PnrF32Saved): ~ (NS * 3648 + 2560) floats per point."""

    def __init__(self, P, NS, device):
        rv, rp = NS * P, P
        f = lambda r, c: torch.empty((r, c), dtype=torch.float32, device=device)  # noqa: E731
        self.P, self.NS = P, NS
        self.in42, self.zlat = f(rv, 64), f(rv, 512)
        self.xin = [f(rv if b < 3 else rp, 512) for b in range(5)]
        self.net = [f(rv if b < 3 else rp, 512) for b in range(5)]
        self.x5 = f(rp, 512)
        self.pool_in = f(rv, 512) if NS > 1 else None
        s = _lib.PnrF32Saved()
        s.in42, s.zlat, s.x5 = self.in42.data_ptr(), self.zlat.data_ptr(), self.x5.data_ptr()
        s.pool_in = self.pool_in.data_ptr() if self.pool_in is not None else None
        for b in range(5):
            s.xin[b], s.net[b] = self.xin[b].data_ptr(), self.net[b].data_ptr()
        self.struct = s

    def release(self):  # interface twin of TrainDumps.release (the fp32 sets are not pooled)
        pass


def eval_ray_samples_f32_train(scene, weights, rays, z, split=False):
    """fp32-precision twin of eval_ray_samples_train: weights = PackedMLP of precision 'f32' (raw nn.Linear tensors).
    split=False: exact fp32 MFMA products (validation grade); True: (head, tail) fp16 operand pairs, 3 f16 MFMAs per product
    (fp32-class, ~6x faster -- precision 'f16x3' under autograd)."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    saved = F32Saved(R * K, scene.NS, rays.device)
    out = torch.empty((R, K, 4), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_eval_ray_samples_f32_train(scene.ref, weights.wref, _p(rays), _p(z), R, max(R // scene.SB, 1), K, _p(out),
                                                      ctypes.byref(saved.struct), int(bool(split)), _stream()), "pnr_eval_ray_samples_f32_train")
    saved.split = bool(split)
    return out, saved


class SplitSaved:
    """What the fused fp32-class training forward keeps (PnrSplitSaved): every wide linear's operand as (head | tail) f16 rows,
    the stream in front of lin_out in fp32, 1-bit relu masks: (NS * 7424 + 6144) bytes per point + masks."""

    def __init__(self, P, NS, device):
        rv, rp = NS * P, P
        h = lambda r, c: torch.empty((2, r, c), dtype=torch.float16, device=device)  # noqa: E731  [head | tail]
        self.P, self.NS = P, NS
        self.in_op, self.zlat = h(rv, 64), h(rv, 512)
        self.a = [h(rv if b < 3 else rp, 512) for b in range(5)]
        self.n = [h(rv if b < 3 else rp, 512) for b in range(5)]
        self.x5 = torch.empty((rp, 512), dtype=torch.float32, device=device)
        self.masks = torch.empty(_lib.load().pnr_train_masks_bytes(P, NS), dtype=torch.uint8, device=device)
        s = _lib.PnrSplitSaved()
        s.in_op, s.zlat, s.x5, s.masks = self.in_op.data_ptr(), self.zlat.data_ptr(), self.x5.data_ptr(), self.masks.data_ptr()
        for b in range(5):
            s.a[b], s.n[b] = self.a[b].data_ptr(), self.n[b].data_ptr()
        self.struct = s

    def release(self):  # interface twin of TrainDumps.release
        pass


def eval_ray_samples_split_train(scene, packed, tables, rays, z):
    """fp32-class training forward through the FUSED split-operand kernel (pnr_eval_ray_samples_split_train): packed = folded
    'f16x3' PackedMLP, tables = fold_latent(scene, state, 'f16x3') of the current parameters.  -> (rgbsigma (R,K,4), SplitSaved)."""
    lib = _lib.load()
    if packed.precision != _lib.PREC_F16X3 or not packed.folded:
        raise ValueError("eval_ray_samples_split_train takes the folded 'f16x3' stream")
    if tables is None:
        raise ValueError("eval_ray_samples_split_train needs the folded lin_z tables (ops.fold_latent)")
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    saved = SplitSaved(R * K, scene.NS, rays.device)
    out = torch.empty((R, K, 4), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_eval_ray_samples_split_train(scene.ref, packed.ptr, _p(tables), _p(rays), _p(z), R, max(R // scene.SB, 1), K,
                                                        _p(out), ctypes.byref(saved.struct), _stream()), "pnr_eval_ray_samples_split_train")
    return out, saved


def mlp_backward_split(weights, saved, g_out, want_d_in=False):
    """Backward of eval_ray_samples_split_train (pnr_mlp_backward_split): weights = PackedMLP of precision 'f32' (the raw
    nn.Linear tensors; the transposed streams are packed inside the call).  -> (grads, d_zlat, d_in | None) like mlp_backward_f32."""
    lib = _lib.load()
    P, NS = saved.P, saved.NS
    g_out = _f32(g_out, "g_out", (P, 4))
    dev = g_out.device
    grads = {k: torch.empty(_MLP_SHAPES.get(k, (512, 512) if k.endswith("weight") else (512,)), dtype=torch.float32, device=dev)
             for k in _MLP_KEYS}
    gstruct, _keep = _weights_struct(grads)
    d_zlat = torch.empty((NS * P, 512), dtype=torch.float32, device=dev)
    d_in = torch.empty((NS * P, 42), dtype=torch.float32, device=dev) if want_d_in else None
    sc = grad_scale(g_out)  # device [s, 1/s]: no host synchronisation
    nbytes = lib.pnr_mlp_backward_split_workspace_bytes(P, NS)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_mlp_backward_split(weights.wref, ctypes.byref(saved.struct), _p(g_out), P, NS, ctypes.byref(gstruct),
                                              _p(d_zlat), _p(d_in), _p(sc), _p(ws), nbytes, _stream()), "pnr_mlp_backward_split")
    return grads, d_zlat, d_in


def mlp_backward_f32(weights, saved, g_out, want_d_in=False):
    """-> ({reference state_dict key: fp32 gradient}, d_zlat (rows_v,512), d_in (rows_v,42) | None) of one ResnetFC: exact fp32
    MFMA products, or the split-operand (fp32-class) form when the forward ran with split=True."""
    lib = _lib.load()
    P, NS = saved.P, saved.NS
    g_out = _f32(g_out, "g_out", (P, 4))
    dev = g_out.device
    grads = {k: torch.empty(_MLP_SHAPES.get(k, (512, 512) if k.endswith("weight") else (512,)), dtype=torch.float32, device=dev)
             for k in _MLP_KEYS}
    gstruct, _keep = _weights_struct(grads)
    d_zlat = torch.empty((NS * P, 512), dtype=torch.float32, device=dev)
    d_in = torch.empty((NS * P, 42), dtype=torch.float32, device=dev) if want_d_in else None
    split = bool(getattr(saved, "split", False))
    sc = grad_scale(g_out) if split else None  # device [s, 1/s]: no host synchronisation
    nbytes = lib.pnr_mlp_backward_f32_workspace_bytes(P, NS)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_mlp_backward_f32(weights.wref, ctypes.byref(saved.struct), _p(g_out), P, NS, ctypes.byref(gstruct),
                                            _p(d_zlat), _p(d_in), int(split), _p(sc), _p(ws), nbytes, _stream()), "pnr_mlp_backward_f32")
    return grads, d_zlat, d_in


def composite_backward(rays, z, rgbsigma, white_bkgd, d_rgb, d_depth=None, d_weights=None, want_dz=False,
                       pre_activation=False):
    """-> dL/d(rgb sigma) per point (R,K,4) [after the output activations, or in front of them with
    pre_activation=True] and optionally dL/dz (R,K)."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    rgbsigma = _f32(rgbsigma, "rgbsigma", (R, K, 4))
    d_rgb = _f32(d_rgb, "d_rgb", (R, 3))
    d_depth = None if d_depth is None else _f32(d_depth, "d_depth", (R,))
    d_weights = None if d_weights is None else _f32(d_weights, "d_weights", (R, K))
    out = torch.empty((R, K, 4), dtype=torch.float32, device=rays.device)
    dz = torch.empty((R, K), dtype=torch.float32, device=rays.device) if want_dz else None
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_composite_backward(_p(rays), _p(z), _p(rgbsigma), R, K, int(bool(white_bkgd)), _p(d_rgb),
                                              _p(d_depth), _p(d_weights), _p(out), _p(dz), int(bool(pre_activation)),
                                              _stream()), "pnr_composite_backward")
    return (out, dz) if want_dz else out


def position_backward(scene, rays, z, d_in42, d_zlat, d_z):
    """accumulate dL/dz through the network inputs into d_z (R,K)."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    rows = scene.NS * R * K
    d_in42 = _f32(d_in42, "d_in42", (rows, 42))
    d_zlat = _f32(d_zlat, "d_zlat", (rows, 512))
    d_z = _f32(d_z, "d_z", (R, K))
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_position_backward(scene.ref, _p(rays), _p(z), R, max(R // scene.SB, 1), K, _p(d_in42),
                                             _p(d_zlat), _p(d_z), _stream()), "pnr_position_backward")
    return d_z


def depth_sample_backward(scene, rays, z, ranks, n4, depth_c, depth_std, d_in42, d_zlat, dz_comp):
    """dL/d(coarse depth) (R,) through the depth samples of the fine pass (nerf.py:157-160,292): network-input term
    (positional code, projection + lookup) + compositing term dz_comp at the samples' sorted positions `ranks`, through
    the clamp; the per-(view, ray, sample) terms are summed in a fixed order (bit-reproducible)."""
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    Kfd = ranks.shape[1]
    if ranks.dtype != torch.int32 or not ranks.is_contiguous() or ranks.shape[0] != R:
        raise ValueError("ranks must be a contiguous (R, Kfd) int32 tensor (ops.sample_fine(..., want_ranks=True))")
    rows = scene.NS * R * K
    n4 = _f32(n4, "n4", (R, Kfd))
    depth_c = _f32(depth_c, "depth_c", (R,))
    d_in42 = _f32(d_in42, "d_in42", (rows, 42))
    d_zlat = _f32(d_zlat, "d_zlat", (rows, 512))
    dz_comp = None if dz_comp is None else _f32(dz_comp, "dz_comp", (R, K))
    contrib = torch.empty((scene.NS, R, Kfd), dtype=torch.float32, device=rays.device)
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_depth_sample_backward(scene.ref, _p(rays), _p(z), R, max(R // scene.SB, 1), K, _p(ranks), _p(n4), Kfd,
                                                 _p(depth_c), float(depth_std), _p(d_in42), _p(d_zlat), _p(dz_comp), _p(contrib),
                                                 _stream()), "pnr_depth_sample_backward")
    return contrib.sum(dim=(0, 2))


def grad_scale(g):
    """-> device tensor (2,): [scale, 1/scale] with scale = 2^(6 - ceil(log2 max|g|)), picked without a host sync
    (NaN when g holds a non-finite value)."""
    lib = _lib.load()
    g = _f32(g, "g")
    out = torch.empty(2, dtype=torch.float32, device=g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.pnr_grad_scale(_p(g), g.numel(), _p(out), _stream()), "pnr_grad_scale")
    return out


def _linear_prec(precision):
    if precision in ("f32", _lib.PREC_F32):
        return _lib.PREC_F32
    if precision in ("f16x3", _lib.PREC_F16X3):
        return _lib.PREC_F16X3
    raise ValueError(f"linear: precision must be 'f16x3' (fp32-class, fast) or 'f32' (exact fp32 products), got {precision!r}")


def linear(x, weight, bias=None, relu_in=False, residual=None, precision="f16x3"):
    """One nn.Linear of ResnetFC / ResnetBlockFC with the ReLU in front and the residual behind folded in
    (src/model/resnetfc.py:53-62,147,175-183):  y = [residual +] [relu](x) weight^T + bias.   x (..., d_in) -> (..., d_out)."""
    lib = _lib.load()
    d_out, d_in = weight.shape
    x2 = _f32(x.reshape(-1, x.shape[-1]), "x", (None, d_in))
    w = _f32(weight, "weight")
    b = None if bias is None else _f32(bias, "bias", (d_out,))
    rows = x2.shape[0]
    r = None if residual is None else _f32(residual.reshape(-1, d_out), "residual", (rows, d_out))
    y = torch.empty((rows, d_out), dtype=torch.float32, device=x2.device)
    with torch.cuda.device(x2.device):
        _lib.check(lib.pnr_linear(_p(x2), _p(w), _p(b), _p(r), _p(y), rows, d_in, d_out, int(bool(relu_in)), _linear_prec(precision),
                                  _stream()), "pnr_linear")
    return y.reshape(*x.shape[:-1], d_out)


def linear_backward(dy, x, weight, relu_in=False, need_dx=True, need_dw=True, need_db=True, precision="f16x3"):
    """-> (dx | None, dweight | None, dbias | None) of `linear` (the residual's gradient is dy itself)."""
    lib = _lib.load()
    d_out, d_in = weight.shape
    x2 = _f32(x.reshape(-1, d_in), "x")
    rows = x2.shape[0]
    g = _f32(dy.reshape(-1, d_out), "dy", (rows, d_out))
    w = _f32(weight, "weight")
    dev = x2.device
    need_dw = need_dw or need_db
    dx = torch.empty((rows, d_in), dtype=torch.float32, device=dev) if need_dx else None
    dw = torch.empty((d_out, d_in), dtype=torch.float32, device=dev) if need_dw else None
    db = torch.empty((d_out,), dtype=torch.float32, device=dev) if need_db else None
    if rows == 0:
        for t in (dx, dw, db):
            if t is not None:
                t.zero_()
        return (None if dx is None else dx.reshape(x.shape)), dw, db
    prec = _linear_prec(precision)
    with torch.cuda.device(dev):
        sc = grad_scale(g) if prec == _lib.PREC_F16X3 else None
        nbytes = lib.pnr_linear_backward_workspace_bytes(d_in, d_out) if need_dw else 0
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if need_dw else None
        _lib.check(lib.pnr_linear_backward(_p(g), _p(x2), _p(w), rows, d_in, d_out, int(bool(relu_in)), _p(dx), _p(dw), _p(db), _p(sc),
                                           _p(ws), nbytes, prec, _stream()), "pnr_linear_backward")
    return (None if dx is None else dx.reshape(x.shape)), dw, db


def mlp_backward(packed_bwd, fwd_dumps, g_out, grad_scale):
    """grad_scale: python float, or a 1-element device tensor (e.g. ops.grad_scale(g_out)[0:1])."""
    lib = _lib.load()
    g_out = _f32(g_out, "g_out", (fwd_dumps.P, 4))
    out = BackwardDumps.acquire(fwd_dumps, g_out.device)
    dev_scale = grad_scale if isinstance(grad_scale, torch.Tensor) else None
    with torch.cuda.device(g_out.device):
        _lib.check(lib.pnr_mlp_backward(packed_bwd.ptr, packed_bwd.precision, ctypes.byref(fwd_dumps.struct), _p(g_out),
                                        1.0 if dev_scale is not None else float(grad_scale), _p(dev_scale), fwd_dumps.P,
                                        fwd_dumps.NS, ctypes.byref(out.struct), _stream()), "pnr_mlp_backward")
    return out


def latent_scatter_single_owner(scene, R, K):
    """True when latent_scatter() on (R rays, K samples) writes every grid element from ONE workgroup (large grids: the tiled form):
    successive calls may then accumulate into one buffer without losing bit-reproducibility (pnr_latent_scatter_single_owner)."""
    return bool(_lib.load().pnr_latent_scatter_single_owner(scene.ref, int(R), max(int(R) // scene.SB, 1), int(K)))


def latent_scatter(scene, rays, z, d_zlat, d_latent_nhwc):
    lib = _lib.load()
    rays = _f32(rays, "rays", (None, 8))
    R = rays.shape[0]
    z = _f32(z, "z", (R, None))
    K = z.shape[1]
    d_zlat = _f32(d_zlat, "d_zlat", (scene.NS * R * K, 512))
    per_obj = max(R // scene.SB, 1)
    # workspace from torch's allocator (projected positions + segment lists of the LDS-slab form): capture-safe, stream-ordered
    nbytes = int(lib.pnr_latent_scatter_workspace_bytes(scene.ref, R, per_obj, K))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=rays.device) if nbytes else None
    with torch.cuda.device(rays.device):
        _lib.check(lib.pnr_latent_scatter(scene.ref, _p(rays), _p(z), R, per_obj, K, _p(d_zlat), _p(d_latent_nhwc),
                                          _p(ws) if ws is not None else None, nbytes, _stream()), "pnr_latent_scatter")
    return d_latent_nhwc


def weight_grad(dY, X, precision, out_scale=1.0, want_bias=True, rows_st=False, cols_st=False):
    """dW (512,512) = out_scale * dY^T X and db (512) = out_scale * sum_rows dY from 16-bit dumps
    (rows,512).  rows_st / cols_st: dY / X are in storage order; results are in feature order."""
    lib = _lib.load()
    rows = dY.shape[0]
    assert dY.shape == (rows, 512) and X.shape == (rows, 512) and dY.dtype == X.dtype and dY.is_cuda
    dY, X = dY.contiguous(), X.contiguous()
    dW = torch.empty((512, 512), dtype=torch.float32, device=dY.device)
    db = torch.empty((512,), dtype=torch.float32, device=dY.device) if want_bias else None
    key = str(dY.device)
    if key not in _wg_workspace:
        _wg_workspace[key] = torch.empty(lib.pnr_weight_grad_workspace_bytes(), dtype=torch.uint8, device=dY.device)
    with torch.cuda.device(dY.device):
        _lib.check(lib.pnr_weight_grad(_p(dY), _p(X), rows, int(precision), float(out_scale), int(bool(rows_st)),
                                       int(bool(cols_st)), _p(dW), _p(db), _p(_wg_workspace[key]), _stream()),
                   "pnr_weight_grad")
    return dW, db


def weight_grad_batched(jobs, precision, out_scale=1.0, out_scale_dev=None):
    """jobs: list of (dY, X, rows_st, cols_st[, x_cols, dw_cols]) with dY (rows,512), X (rows,x_cols=512) 16-bit
    dumps -> list of (dW (512,dw_cols), db (512)), all computed by ONE pnr_weight_grad_batched call (<= 16 jobs).
    precision 'f16x3' (split-operand, fp32-class): dY (2,rows,512), X (2,rows,x_cols) float16 = [head | tail] row sets."""
    lib = _lib.load()
    n = len(jobs)
    dev = jobs[0][0].device
    arr = (_lib.PnrWeightGradJob * n)()
    outs, keep = [], []
    max_rows = 0
    if n > 16:
        raise _lib.PixelNerfHipError("weight_grad_batched: at most 16 jobs per call")
    for j, job in enumerate(jobs):
        dY, X, rows_st, cols_st = job[:4]
        x_cols, dw_cols = (job[4], job[5]) if len(job) > 4 else (512, 512)
        if int(precision) == _lib.PREC_F16X3:
            rows = dY.shape[1]
            assert dY.shape == (2, rows, 512) and X.shape == (2, rows, x_cols) and dY.dtype == X.dtype == torch.float16 and dY.is_cuda
        else:
            rows = dY.shape[0]
            assert dY.shape == (rows, 512) and X.shape == (rows, x_cols) and dY.dtype == X.dtype and dY.is_cuda
        dY, X = dY.contiguous(), X.contiguous()
        dW = torch.empty((512, dw_cols), dtype=torch.float32, device=dev)
        db = torch.empty((512,), dtype=torch.float32, device=dev)
        keep.append((dY, X))
        outs.append((dW, db))
        arr[j].dY, arr[j].X, arr[j].rows = dY.data_ptr(), X.data_ptr(), rows
        arr[j].rows_storage_order, arr[j].cols_storage_order = int(bool(rows_st)), int(bool(cols_st))
        arr[j].dW, arr[j].db = dW.data_ptr(), db.data_ptr()
        arr[j].x_cols, arr[j].dw_cols = x_cols, dw_cols
        max_rows = max(max_rows, rows)
    need = lib.pnr_weight_grad_batched_workspace_bytes(n, max_rows)
    key = ("batched", str(dev))
    if key not in _wg_workspace or _wg_workspace[key].numel() < need:
        _wg_workspace[key] = torch.empty(need, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_weight_grad_batched(arr, n, int(precision), float(out_scale), _p(out_scale_dev),
                                               _p(_wg_workspace[key]), _stream()),
                   "pnr_weight_grad_batched")
    return outs


def lin_out_grad(g_out, x5, precision):
    """lin_out: dW (4,512) = g_out^T x5 (feature order), db (4) = column sums; g_out (P,4) fp32, x5 (P,512) dump."""
    lib = _lib.load()
    g_out = _f32(g_out, "g_out", (None, 4))
    P = g_out.shape[0]
    assert x5.shape == (P, 512) and x5.is_cuda
    x5 = x5.contiguous()
    dev = g_out.device
    dW = torch.empty((4, 512), dtype=torch.float32, device=dev)
    db = torch.empty((4,), dtype=torch.float32, device=dev)
    key = ("lin_out", str(dev))
    if key not in _wg_workspace:
        _wg_workspace[key] = torch.empty(lib.pnr_lin_out_grad_workspace_bytes(), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.pnr_lin_out_grad(_p(g_out), _p(x5), P, int(precision), _p(dW), _p(db), _p(_wg_workspace[key]), _stream()),
                   "pnr_lin_out_grad")
    return dW, db


_wg_workspace = {}
