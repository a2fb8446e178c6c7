"""
ctypes binding of libpixelnerf_hip.so (C ABI: include/pixelnerf_hip.h).

The shared library is built in-tree (pixel-nerf_amd/csrc/libpixelnerf_hip.so) by
`build_library()` (called from __graft_entry__.build()); there is NO fallback: if the library
is missing or fails to load, every product entry point raises.
"""
import ctypes
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
# PIXELNERF_HIP_LIB selects another build of the library (A/B experiments, tools/build_variant.sh).  A build made with ANY
# experiment switch is compiled with -DPNR_VARIANT and reports a NEGATIVE ABI revision: load() refuses it unless the
# process says PIXELNERF_ALLOW_VARIANT=1 (the A/B tools do) -- a stray -D can no longer yield a library that passes for the product
LIB_PATH = os.environ.get("PIXELNERF_HIP_LIB") or os.path.join(CSRC, "libpixelnerf_hip.so")
SOURCES = ["pnr_api.hip", "pnr_pack.hip", "pnr_render.hip", "pnr_mlp.hip", "pnr_split.hip", "pnr_bwd.hip", "pnr_f32.hip", "pnr_encode.hip"]
HEADERS = ["pnr_common.h", "pnr_layout.h", "pnr_device.h", "pnr_raysrc.h", "pnr_internal.h", os.path.join("..", "..", "include", "pixelnerf_hip.h")]

ABI_VERSION = 8  # PNR_ABI_VERSION of the include/pixelnerf_hip.h this binding (struct layouts, argtypes below) was written against
PREC_F16, PREC_BF16, PREC_F32, PREC_F16X3 = 0, 1, 2, 3
PRECISIONS = {"f16": PREC_F16, "fp16": PREC_F16, "bf16": PREC_BF16, "f32": PREC_F32, "fp32": PREC_F32, "f16x3": PREC_F16X3}

c_float_p = ctypes.c_void_p  # device pointers travel as plain addresses


class PnrScene(ctypes.Structure):
    _fields_ = [
        ("latent_nhwc", ctypes.c_void_p), ("poses", ctypes.c_void_p), ("focal", ctypes.c_void_p),
        ("c", ctypes.c_void_p),
        ("SB", ctypes.c_int32), ("NS", ctypes.c_int32), ("Hl", ctypes.c_int32), ("Wl", ctypes.c_int32),
        ("n_focal", ctypes.c_int32), ("n_c", ctypes.c_int32),
        ("img_w", ctypes.c_float), ("img_h", ctypes.c_float),
    ]


class PnrMlpWeights(ctypes.Structure):
    _fields_ = [
        ("lin_in_w", ctypes.c_void_p), ("lin_in_b", ctypes.c_void_p),
        ("lin_z_w", ctypes.c_void_p * 3), ("lin_z_b", ctypes.c_void_p * 3),
        ("fc0_w", ctypes.c_void_p * 5), ("fc0_b", ctypes.c_void_p * 5),
        ("fc1_w", ctypes.c_void_p * 5), ("fc1_b", ctypes.c_void_p * 5),
        ("lin_out_w", ctypes.c_void_p), ("lin_out_b", ctypes.c_void_p),
        ("combine_max", ctypes.c_int32),
    ]


class PnrWeightGradJob(ctypes.Structure):
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("rows", ctypes.c_longlong),
                ("rows_storage_order", ctypes.c_int), ("cols_storage_order", ctypes.c_int),
                ("dW", ctypes.c_void_p), ("db", ctypes.c_void_p), ("x_cols", ctypes.c_int), ("dw_cols", ctypes.c_int)]


class PnrTrainDumps(ctypes.Structure):
    _fields_ = [("d_in", ctypes.c_void_p), ("d_z", ctypes.c_void_p), ("d_a", ctypes.c_void_p * 5),
                ("d_n", ctypes.c_void_p * 5), ("d_x5", ctypes.c_void_p), ("d_mask", ctypes.c_void_p)]


class PnrBackwardDumps(ctypes.Structure):
    _fields_ = [("g_fc1", ctypes.c_void_p * 5), ("g_fc0", ctypes.c_void_p * 5), ("g_x0", ctypes.c_void_p),
                ("d_zlat", ctypes.c_void_p), ("d_in", ctypes.c_void_p)]


class PnrF32Saved(ctypes.Structure):
    _fields_ = [("in42", ctypes.c_void_p), ("zlat", ctypes.c_void_p), ("xin", ctypes.c_void_p * 5), ("net", ctypes.c_void_p * 5),
                ("x5", ctypes.c_void_p), ("pool_in", ctypes.c_void_p)]


class PnrSplitSaved(ctypes.Structure):
    _fields_ = [("in_op", ctypes.c_void_p), ("zlat", ctypes.c_void_p), ("a", ctypes.c_void_p * 5), ("n", ctypes.c_void_p * 5),
                ("x5", ctypes.c_void_p), ("masks", ctypes.c_void_p)]


# every symbol include/pixelnerf_hip.h declares: name -> (restype, argtypes)
_I, _F, _P, _SZ = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
PROTOTYPES = {
    "pnr_last_error": (ctypes.c_char_p, []),
    "pnr_version": (_I, [ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "pnr_abi_version": (_I, []),
    "pnr_device_info": (_I, [ctypes.POINTER(_I), ctypes.POINTER(_I)]),
    "pnr_params_checksum_ws_bytes": (_SZ, []),
    "pnr_params_checksum": (_I, [ctypes.POINTER(PnrMlpWeights), _P, _P, _P, _P, _P]),
    "pnr_packed_mlp_bytes": (_SZ, []),
    "pnr_pack_mlp": (_I, [ctypes.POINTER(PnrMlpWeights), _I, _P, _P]),
    "pnr_nchw_to_nhwc": (_I, [_P, _P, _I, _I, _I, _I, _P]),
    "pnr_sample_coarse": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "pnr_sample_fine": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P]),
    "pnr_eval_ray_samples": (_I, [ctypes.POINTER(PnrScene), _P, _I, _P, _P, _I, _I, _I, _P, _P]),
    "pnr_eval_points": (_I, [ctypes.POINTER(PnrScene), _P, _I, _P, _P, _I, _P, _P]),
    "pnr_render_views_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I]),
    "pnr_render_views": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _P, _I, _I, _I, _F, _F, _F, _F, _F, _F, _I, _I, _I, _F,
                              _I, _I, _P, _P, _P, _P, ctypes.c_ulonglong, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pnr_philox_noise": (_I, [ctypes.c_ulonglong, ctypes.c_longlong, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "pnr_philox_raw": (_I, [ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "pnr_render_forward_seeded": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _I,
                                       ctypes.c_ulonglong, ctypes.c_longlong, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pnr_pyramid_to_latent": (_I, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                   ctypes.POINTER(ctypes.c_int), _I, _I, _P, _P, _P]),
    "pnr_saturation_guard": (_I, [_P]),
    "pnr_grid_index": (_I, [_P, _I, _I, _I, _I, _P, ctypes.c_longlong, _P, _P]),
    "pnr_grid_index_backward": (_I, [_P, _I, _I, _I, _I, _P, ctypes.c_longlong, _P, _P, _P, _P]),
    "pnr_positional_encoding": (_I, [_P, ctypes.c_longlong, _I, _I, _P, _P, _I, _P, _P]),
    "pnr_positional_encoding_backward": (_I, [_P, _P, ctypes.c_longlong, _I, _I, _P, _P, _I, _P, _P]),
    "pnr_sample_training_rays": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    "pnr_eval_epilogue": (_I, [_P, _P, _I, _I, _F, _F, _P, _P, _P, _P, _P, _P]),
    "pnr_resnetfc_forward_f32_workspace_bytes": (_SZ, [ctypes.c_longlong, _I]),
    "pnr_resnetfc_forward_f32": (_I, [ctypes.POINTER(PnrMlpWeights), _P, ctypes.c_longlong, _I, _I, _P, _P, _SZ, _P]),
    "pnr_train_masks_bytes": (_SZ, [ctypes.c_longlong, _I]),
    "pnr_eval_f32_workspace_bytes": (_SZ, [_I, ctypes.c_longlong]),
    "pnr_eval_ray_samples_f32": (_I, [ctypes.POINTER(PnrScene), ctypes.POINTER(PnrMlpWeights), _P, _P, _I, _I, _I, _P,
                                      _P, _SZ, _P]),
    "pnr_eval_points_f32": (_I, [ctypes.POINTER(PnrScene), ctypes.POINTER(PnrMlpWeights), _P, _P, _I, _P, _P, _SZ, _P]),
    "pnr_folded_tables_bytes": (_SZ, [ctypes.POINTER(PnrScene)]),
    "pnr_fold_latent": (_I, [ctypes.POINTER(PnrScene), ctypes.POINTER(PnrMlpWeights), _I, _P, _P]),
    "pnr_pack_mlp_folded": (_I, [ctypes.POINTER(PnrMlpWeights), _I, _P, _P]),
    "pnr_eval_ray_samples_folded": (_I, [ctypes.POINTER(PnrScene), _P, _P, _I, _P, _P, _I, _I, _I, _P, _P]),
    "pnr_eval_points_folded": (_I, [ctypes.POINTER(PnrScene), _P, _P, _I, _P, _P, _I, _P, _P]),
    "pnr_render_forward_folded": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _I,
                                       _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pnr_packed_mlp_split_bytes": (_SZ, []),
    "pnr_pack_mlp_split": (_I, [ctypes.POINTER(PnrMlpWeights), _P, _P]),
    "pnr_folded_tables_f32_bytes": (_SZ, [ctypes.POINTER(PnrScene)]),
    "pnr_fold_latent_f32": (_I, [ctypes.POINTER(PnrScene), ctypes.POINTER(PnrMlpWeights), _P, _P]),
    "pnr_eval_ray_samples_split": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "pnr_eval_points_split": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _P, _P]),
    "pnr_eval_ray_samples_train": (_I, [ctypes.POINTER(PnrScene), _P, _I, _P, _P, _I, _I, _I, _P,
                                        ctypes.POINTER(PnrTrainDumps), _P]),
    "pnr_storage_perm": (_I, [ctypes.POINTER(ctypes.c_int32)]),
    "pnr_packed_mlp_bwd_bytes": (_SZ, []),
    "pnr_pack_mlp_bwd": (_I, [ctypes.POINTER(PnrMlpWeights), _I, _P, _P]),
    "pnr_composite_backward": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _I, _P]),
    "pnr_position_backward": (_I, [ctypes.POINTER(PnrScene), _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "pnr_depth_sample_backward": (_I, [ctypes.POINTER(PnrScene), _P, _P, _I, _I, _I, _P, _P, _I, _P, _F, _P, _P, _P, _P, _P]),
    "pnr_mlp_backward": (_I, [_P, _I, ctypes.POINTER(PnrTrainDumps), _P, _F, _P, ctypes.c_longlong, _I,
                              ctypes.POINTER(PnrBackwardDumps), _P]),
    "pnr_weight_grad_workspace_bytes": (_SZ, []),
    "pnr_weight_grad": (_I, [_P, _P, ctypes.c_longlong, _I, _F, _I, _I, _P, _P, _P, _P]),
    "pnr_lin_out_grad_workspace_bytes": (_SZ, []),
    "pnr_lin_out_grad": (_I, [_P, _P, ctypes.c_longlong, _I, _P, _P, _P, _P]),
    "pnr_weight_grad_batched_workspace_bytes": (_SZ, [_I, ctypes.c_longlong]),
    "pnr_weight_grad_batched": (_I, [ctypes.POINTER(PnrWeightGradJob), _I, _I, _F, _P, _P, _P]),
    "pnr_grad_scale": (_I, [_P, ctypes.c_longlong, _P, _P]),
    "pnr_linear": (_I, [_P, _P, _P, _P, _P, ctypes.c_longlong, _I, _I, _I, _I, _P]),
    "pnr_linear_backward_workspace_bytes": (_SZ, [_I, _I]),
    "pnr_linear_backward": (_I, [_P, _P, _P, ctypes.c_longlong, _I, _I, _I, _P, _P, _P, _P, _P, _SZ, _I, _P]),
    "pnr_fold_latent_f32_rows_workspace_bytes": (_SZ, [ctypes.POINTER(PnrScene)]),
    "pnr_fold_latent_f32_rows": (_I, [ctypes.POINTER(PnrScene), ctypes.POINTER(PnrMlpWeights), _P, _P, _I, _I, _I, _P, _P, _SZ, _P]),
    "pnr_latent_scatter_workspace_bytes": (_SZ, [ctypes.POINTER(PnrScene), _I, _I, _I]),
    "pnr_latent_scatter_single_owner": (_I, [ctypes.POINTER(PnrScene), _I, _I, _I]),
    "pnr_latent_scatter": (_I, [ctypes.POINTER(PnrScene), _P, _P, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "pnr_composite": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "pnr_render_workspace_bytes": (_SZ, [_I, _I, _I]),
    "pnr_render_forward": (_I, [ctypes.POINTER(PnrScene), _P, _P, _I, _P, _I, _I, _I, _I, _I, _F, _I, _I,
                                _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pnr_gen_rays": (_I, [_P, _I, _I, _I, _F, _F, _F, _F, _F, _F, _P, _P]),
    "pnr_eval_ray_samples_split_train": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _I, _I, _P, ctypes.POINTER(PnrSplitSaved), _P]),
    "pnr_eval_ray_samples_f32_train": (_I, [ctypes.POINTER(PnrScene), ctypes.POINTER(PnrMlpWeights), _P, _P, _I, _I, _I, _P,
                                            ctypes.POINTER(PnrF32Saved), _I, _P]),
    "pnr_mlp_backward_f32_workspace_bytes": (_SZ, [ctypes.c_longlong, _I]),
    "pnr_mlp_backward_f32": (_I, [ctypes.POINTER(PnrMlpWeights), ctypes.POINTER(PnrF32Saved), _P, ctypes.c_longlong, _I,
                                  ctypes.POINTER(PnrMlpWeights), _P, _P, _I, _P, _P, _SZ, _P]),
    "pnr_mlp_backward_split_workspace_bytes": (_SZ, [ctypes.c_longlong, _I]),
    "pnr_mlp_backward_split": (_I, [ctypes.POINTER(PnrMlpWeights), ctypes.POINTER(PnrSplitSaved), _P, ctypes.c_longlong, _I,
                                    ctypes.POINTER(PnrMlpWeights), _P, _P, _P, _P, _SZ, _P]),
    "pnr_point_features_f32": (_I, [ctypes.POINTER(PnrScene), _P, _P, _I, _P, _P, _P]),
    "pnr_profile_enable": (_I, [_I]),
    "pnr_profile_read": (_I, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I)]),
}
# test hook exported by the library but not part of the public header
_EXTRA = {"pnr_debug_set_x_dump": (_I, [_P]),
          "pnr_debug_phase_timing": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _I, _I, _P, _P]),
          "pnr_debug_phase_timing_split": (_I, [ctypes.POINTER(PnrScene), _P, _P, _P, _P, _I, _I, _I, _P, _P, _P])}

_lib = None


class PixelNerfHipError(RuntimeError):
    pass


def _needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_library(force=False, verbose=False, jobs=None):
    """Compile the HIP sources for gfx950 into csrc/libpixelnerf_hip.so (hipcc cross-compiles
    without a GPU).  No-op when the library is newer than every source.  One hipcc per translation unit, in parallel, objects
    kept under build/obj_prod/ (a unit is recompiled when its source or any header is newer than its object), then one link."""
    if not force and not _needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
    objdir = os.path.join(os.path.dirname(os.path.dirname(CSRC)), "build", "obj_prod")
    os.makedirs(objdir, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))

    def compile_unit(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        path = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), newest_header):
            return obj, None
        cmd = [hipcc] + flags + ["-c", path, "-o", obj + f".tmp.{os.getpid()}"]
        if verbose:
            print(" ".join(cmd[:-1] + [obj]), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            return obj, res.stdout + res.stderr
        os.replace(cmd[-1], obj)
        return obj, None

    with ThreadPoolExecutor(max_workers=jobs or min(len(SOURCES), os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_unit, SOURCES))
    errors = [e for _, e in results if e]
    if errors:
        raise PixelNerfHipError("hipcc failed:\n" + "\n".join(errors))
    tmp = f"{LIB_PATH}.tmp.{os.getpid()}"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _ in results] + ["-o", tmp]
    if verbose:
        print(" ".join(cmd).replace(tmp, LIB_PATH), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise PixelNerfHipError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)  # atomic: a process that already mapped the old file keeps it
    return LIB_PATH


def ensure_built():
    """Build the library if it is missing or older than its sources, safely under concurrent
    callers (the ranks of a torch.distributed.run launch): one builds under an exclusive file
    lock, the others wait and find it done.  For entry scripts (bench.py, smoke, the test
    session); load() itself never builds."""
    import fcntl
    if not _needs_build():
        return LIB_PATH
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return build_library(force=False)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def load():
    """dlopen the library and bind every prototype; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, same SONAME as
    # /opt/rocm's).  Streams and device pointers are shared with torch, so torch's copy must be
    # the one in the process: import torch BEFORE dlopen so our DT_NEEDED resolves to it.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PixelNerfHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(pixelnerf_amd has no non-HIP fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    # struct layouts and argument lists are positional: a library built from another revision of the header (a stale
    # build, an A/B variant selected with PIXELNERF_HIP_LIB) would bind silently and read fields at wrong offsets
    try:
        lib.pnr_abi_version.restype = ctypes.c_int
        abi = lib.pnr_abi_version()
    except AttributeError:
        abi = None
    if abi is not None and abi < 0 and os.environ.get("PIXELNERF_ALLOW_VARIANT") == "1":
        abi = -abi  # an experiment build, asked for explicitly
    if abi != ABI_VERSION:
        raise PixelNerfHipError(f"{LIB_PATH} implements ABI revision {abi}, this binding was written against {ABI_VERSION} "
                                "(include/pixelnerf_hip.h PNR_ABI_VERSION): rebuild it (__graft_entry__.build())")
    for name, (res, args) in list(PROTOTYPES.items()) + list(_EXTRA.items()):
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pnr_last_error()
        raise PixelNerfHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")
