"""CPU tests: the C-ABI library builds for gfx950, loads, and exports every symbol that
include/pixelnerf_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from pixelnerf_amd import _lib


@pytest.fixture(scope="module")
def lib():
    _lib.build_library()
    return _lib.load()


def header_functions(repo_root):
    src = open(os.path.join(repo_root, "include", "pixelnerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pnr_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib, repo_root):
    names = header_functions(repo_root)
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pixelnerf_hip.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes and header out of sync"


def test_struct_layouts_match_header():
    # PnrScene: 4 pointers, 6 int32, 2 floats; PnrMlpWeights: 30 pointers + the combine_max flag (padded to 8); dumps: 13 / 13 pointers (11 dY dumps + d_zlat + d_in)
    assert ctypes.sizeof(_lib.PnrTrainDumps) == 14 * 8 and ctypes.sizeof(_lib.PnrBackwardDumps) == 13 * 8
    assert ctypes.sizeof(_lib.PnrScene) == 4 * 8 + 6 * 4 + 2 * 4
    assert ctypes.sizeof(_lib.PnrMlpWeights) == 30 * 8 + 8


def test_struct_layouts_against_the_header_compiled_by_gcc(repo_root, tmp_path):
    """every struct of include/pixelnerf_hip.h: sizeof and the offset of every field as gcc lays the header out, against the ctypes
    mirror in pixelnerf_amd/_lib.py (the header is plain C: a reference maintainer's cgo / cffi binding sees the same layout)"""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    structs = {"PnrScene": _lib.PnrScene, "PnrMlpWeights": _lib.PnrMlpWeights, "PnrTrainDumps": _lib.PnrTrainDumps,
               "PnrBackwardDumps": _lib.PnrBackwardDumps, "PnrF32Saved": _lib.PnrF32Saved, "PnrSplitSaved": _lib.PnrSplitSaved,
               "PnrWeightGradJob": _lib.PnrWeightGradJob}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pixelnerf_hip.h"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu\\n", sizeof({name}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{name}.{fname} %zu\\n", offsetof({name}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(repo_root, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, cls in structs.items():
        assert int(got[name]) == ctypes.sizeof(cls), name
        for fname, _ in cls._fields_:
            assert int(got[f"{name}.{fname}"]) == getattr(cls, fname).offset, f"{name}.{fname}"


def test_abi_revision_header_library_binding_agree(lib, repo_root):
    """include/pixelnerf_hip.h, the built library and the ctypes binding carry the same ABI revision; load() refuses
    a library of another revision (a stale build or an A/B variant) instead of binding structs at wrong offsets."""
    src = open(os.path.join(repo_root, "include", "pixelnerf_hip.h")).read()
    hdr = int(re.search(r"#define\s+PNR_ABI_VERSION\s+(\d+)", src).group(1))
    assert hdr == _lib.ABI_VERSION == lib.pnr_abi_version()
    saved, saved_lib = _lib.ABI_VERSION, _lib._lib
    try:
        _lib.ABI_VERSION, _lib._lib = saved + 1, None
        with pytest.raises(_lib.PixelNerfHipError, match="ABI revision"):
            _lib.load()
    finally:
        _lib.ABI_VERSION, _lib._lib = saved, saved_lib


def test_host_only_entry_points(lib):
    major, minor = ctypes.c_int(-1), ctypes.c_int(-1)
    assert lib.pnr_version(ctypes.byref(major), ctypes.byref(minor)) == 0
    assert (major.value, minor.value) == (0, 1)
    # packed stream: 8 waves x 424 ring steps x 2 fragments x 1 KiB + biases + (b_out, network flags word, pad)
    assert lib.pnr_packed_mlp_bytes() == 8 * 424 * 2 * 1024 + 11 * 8 * 64 * 4 + 32
    assert lib.pnr_render_workspace_bytes(0, 64, 128) == 0
    # backward stream: head 132 + per view (6 block GEMMs + 3 lin_z^T) x 32 + lin_in^T 4 = 424 ring steps
    assert lib.pnr_packed_mlp_bwd_bytes() == 8 * 424 * 2 * 1024
    assert lib.pnr_packed_mlp_split_bytes() == 2 * lib.pnr_packed_mlp_bytes()
    assert lib.pnr_weight_grad_workspace_bytes() == 32 * (512 * 512 + 512) * 4
    perm = (ctypes.c_int32 * 512)()
    assert lib.pnr_storage_perm(perm) == 0 and sorted(perm) == list(range(512)) and perm[16] == 4 and perm[1] == 1
    r, kc, kf = 100, 64, 128
    fl = lambda n: (n + 63) // 64 * 64
    expect = 4 * (fl(r * kc) + fl(r * kc * 4) + fl(r * kc) + fl(r * (kc + kf)) + fl(r * (kc + kf) * 4)
                  + fl(r * kf) + fl(r * kf * 4) + fl(r * (kc + kf)))  # + new samples / their outputs / ranks (coarse-network reuse)
    assert lib.pnr_render_workspace_bytes(r, kc, kf) == expect


def test_argument_validation_without_gpu(lib):
    # invalid arguments are rejected on the host before any HIP call
    assert lib.pnr_sample_coarse(None, None, 4, 0, 0, None, None) == -1
    assert b"bad sizes" in lib.pnr_last_error()
    assert lib.pnr_sample_fine(None, None, None, None, None, None, None, 4, 6000, 0, 0, 0.01, 0, None, None, None) == -1
    assert b"samples per ray" in lib.pnr_last_error()  # the ray's cdf + sample set must fit the LDS (no such limit below ~10 000)
    assert lib.pnr_composite(None, None, None, 3, 8, 0, None, None, None, None) == -1
    assert lib.pnr_pack_mlp(None, 0, None, None) == -1
    assert lib.pnr_pack_mlp_bwd(None, 0, None, None) == -1
    assert lib.pnr_composite_backward(None, None, None, 3, 8, 0, None, None, None, None, None, 0, None) == -1
    assert lib.pnr_mlp_backward(None, 0, None, None, 1.0, None, 10, 1, None, None) == -1
    # d_zlat is required: the chain's weight ring only stays in step with the transposed stream when the lin_z^T GEMMs run
    dumps, bd = _lib.PnrTrainDumps(), _lib.PnrBackwardDumps()
    dumps.d_mask = 64
    for b in range(5):
        bd.g_fc1[b], bd.g_fc0[b] = 64, 64
    bd.g_x0 = 64
    assert lib.pnr_mlp_backward(64, 0, ctypes.byref(dumps), 64, 1.0, None, 10, 1, ctypes.byref(bd), None) == -1
    assert b"d_zlat is required" in lib.pnr_last_error()
    assert lib.pnr_grad_scale(None, 10, None, None) == -1
    assert lib.pnr_weight_grad(None, None, 10, 0, 1.0, 0, 0, None, None, None, None) == -1
    assert lib.pnr_position_backward(None, None, None, 1, 1, 1, None, None, None, None) == -1
    # round-6 entries (ABI rev 8): the row-wise fold of training passes on large grids, the scatter's ownership query
    sc = _lib.PnrScene()
    sc.SB, sc.NS, sc.Hl, sc.Wl = 2, 3, 150, 200
    M = 2 * 3 * 150 * 200
    nb = (M + 4095) // 4096
    assert lib.pnr_fold_latent_f32_rows_workspace_bytes(ctypes.byref(sc)) == nb * 4096 + (nb + 4 + M) * 4
    assert lib.pnr_fold_latent_f32_rows_workspace_bytes(None) == 0
    assert lib.pnr_fold_latent_f32_rows(None, None, None, None, 4, 2, 8, None, None, 0, None) == -1
    assert b"null argument" in lib.pnr_last_error()
    assert lib.pnr_latent_scatter_single_owner(None, 4, 2, 8) == 0
    assert lib.pnr_latent_scatter_single_owner(ctypes.byref(sc), 256, 128, 96) == 1   # DTU-sized grid: owner tiles
    sc.Hl, sc.Wl = 32, 32
    assert lib.pnr_latent_scatter_single_owner(ctypes.byref(sc), 256, 128, 96) == 0   # LDS slabs, two workgroups per (image, slice)
    # the stand-alone nn.Linear operator pair: sizes, precisions (exact fp32 / fp32-class only), operands, workspace, grad_scale
    assert lib.pnr_linear(None, None, None, None, None, 4, 0, 8, 0, _lib.PREC_F32, None) == -1
    assert lib.pnr_linear(64, 64, None, None, 64, 4, 8, 8, 0, _lib.PREC_F16, None) == -1
    assert b"PNR_PREC_F32 or PNR_PREC_F16X3" in lib.pnr_last_error()
    assert lib.pnr_linear(None, 64, None, None, 64, 4, 8, 8, 0, _lib.PREC_F32, None) == -1
    assert lib.pnr_linear(None, None, None, None, None, 0, 8, 8, 0, _lib.PREC_F32, None) == 0   # no rows: a no-op
    assert lib.pnr_linear_backward_workspace_bytes(42, 512) == 32 * (42 * 512 + 512) * 4 and lib.pnr_linear_backward_workspace_bytes(0, 4) == 0
    assert lib.pnr_linear_backward(64, 64, 64, 4, 8, 8, 0, None, 64, None, None, None, 0, _lib.PREC_F32, None) == -1
    assert b"workspace too small" in lib.pnr_last_error()
    assert lib.pnr_linear_backward(64, 64, 64, 4, 8, 8, 0, 64, None, None, None, None, 0, _lib.PREC_F16X3, None) == -1
    assert b"grad_scale" in lib.pnr_last_error()
    assert lib.pnr_linear_backward(64, 64, 64, 4, 8, 8, 0, None, None, 64, None, None, 0, _lib.PREC_F32, None) == -1  # db without dW
    # empty batches are a successful no-op (reference: empty output, nerf.py:23-27)
    assert lib.pnr_sample_coarse(None, None, 0, 8, 0, None, None) == 0
    assert lib.pnr_composite(None, None, None, 0, 8, 0, None, None, None, None) == 0


def test_ops_refuse_cpu_tensors():
    import torch
    from pixelnerf_amd import ops
    with pytest.raises(_lib.PixelNerfHipError):
        ops.sample_coarse(torch.zeros(2, 8), torch.zeros(2, 4))
    with pytest.raises(_lib.PixelNerfHipError):
        ops.composite(torch.zeros(2, 8), torch.zeros(2, 4), torch.zeros(2, 4, 4))
