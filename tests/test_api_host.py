"""CPU tests of the host-side mirror of the reference API: construction, attributes,
state_dict compatibility, config shim, output containers, loud failure without a GPU."""
import os

import pytest
import torch

from pixelnerf_amd import _lib
from pixelnerf_amd.model import make_model
from pixelnerf_amd.model.code import PositionalEncoding
from pixelnerf_amd.render import NeRFRenderer
from pixelnerf_amd.render.nerf import _RenderWrapper
from pixelnerf_amd.util import Conf, DotMap, combine_interleaved, gen_rays, psnr, repeat_interleave
from pixelnerf_amd.util.conf import default_model_conf, default_renderer_conf

from helpers import GOLDEN_DIR, load_golden


@pytest.fixture(scope="module")
def net():
    torch.manual_seed(0)
    return make_model(default_model_conf())


def test_state_dict_matches_reference_manifest(net):
    """every non-encoder key/shape of the reference checkpoint exists here (reference
    checkpoints must load: SURVEY.md §5)."""
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    rend = {k: tuple(v.shape) for k, v in NeRFRenderer(n_coarse=64, n_fine=32).state_dict().items()}
    n = 0
    for line in open(os.path.join(GOLDEN_DIR, "state_dict_manifest.txt")):
        kind, key, shape = line.split()
        shape = () if shape == "scalar" else tuple(int(s) for s in shape.split("x"))
        src = ours if kind == "net" else rend
        assert key in src, f"missing {kind} key {key}"
        assert src[key] == shape, (key, src[key], shape)
        n += 1
    assert n == 64
    non_enc = [k for k in ours if not k.startswith("encoder.")]
    assert len(non_enc) == 62  # nothing extra either
    # torchvision resnet34 naming for the encoder (incl. the unused layer4)
    for k in ("encoder.model.conv1.weight", "encoder.model.bn1.running_mean",
              "encoder.model.layer1.2.conv2.weight", "encoder.model.layer2.0.downsample.0.weight",
              "encoder.model.layer3.5.bn2.bias", "encoder.model.layer4.2.conv1.weight"):
        assert k in ours, k
    assert ours["encoder.model.layer2.0.downsample.0.weight"] == (128, 64, 1, 1)


def test_mlp_init_follows_reference(net):
    # resnetfc.py:36-39: fc_1 zero-initialised, biases zero, others kaiming fan-in
    blk = net.mlp_coarse.blocks[0]
    assert blk.fc_1.weight.abs().max() == 0 and blk.fc_0.bias.abs().max() == 0
    std = net.mlp_coarse.lin_z[0].weight.std().item()
    assert abs(std - (2.0 / 512) ** 0.5) < 5e-3
    assert net.mlp_coarse.supported() and net.d_in == 42 and net.d_latent == 512
    assert sum(p.numel() for p in net.mlp_coarse.parameters()) == 3438596  # SURVEY.md headline facts


def test_positional_encoding_module_keeps_the_checkpoint_contract():
    """`_freqs` / `_phases` are what a reference checkpoint holds (src/model/code.py:17-28, restated here); the forward is a
    HIP operator (tests/test_hip_features.py) and refuses CPU tensors instead of falling back."""
    import numpy as np
    from pixelnerf_amd._lib import PixelNerfHipError
    code = PositionalEncoding(num_freqs=6, d_in=3, freq_factor=1.5, include_input=True)
    ref_freqs = torch.repeat_interleave(1.5 * 2.0 ** torch.arange(0, 6), 2).view(1, -1, 1)
    ref_phases = torch.zeros(12)
    ref_phases[1::2] = np.pi * 0.5
    sd = code.state_dict()
    assert list(sd) == ["_freqs", "_phases"]
    assert sd["_freqs"].dtype == torch.float32 and torch.equal(sd["_freqs"], ref_freqs)
    assert torch.equal(sd["_phases"], ref_phases.view(1, -1, 1))
    assert code.d_out == 39 and PositionalEncoding(4, 2, include_input=False).d_out == 16
    with pytest.raises(PixelNerfHipError):
        code(torch.from_numpy(load_golden("stages")["posenc_x"]))


def test_encode_state_conventions(net):
    """encode() leaves the reference's buffers (models.py:111-141); checked against the oracle's
    synthetic encode_state on CPU (the encoder trunk itself is plain PyTorch)."""
    from testdata import synthetic
    n2 = make_model(default_model_conf()).eval()
    src = torch.stack([synthetic.pose_spherical(30.0, -20.0, 2.7), synthetic.pose_spherical(80.0, -10.0, 2.7)])
    imgs = torch.rand(1, 2, 3, 64, 64) * 2 - 1
    with torch.no_grad():
        n2.encode(imgs, src[None], torch.tensor(119.4), c=None)
    assert n2.num_objs == 1 and n2.num_views_per_obj == 2
    assert tuple(n2.encoder.latent.shape) == (2, 512, 32, 32)  # conv1 stride 2; sn64 would skip the pool
    poses, focal, c, ishape = synthetic.encode_state(src, (119.4, 119.4), (32.0, 32.0), 64, 64)
    assert torch.allclose(n2.poses, poses, atol=1e-6)
    assert torch.allclose(n2.focal, focal) and torch.allclose(n2.c, c) and torch.allclose(n2.image_shape, ishape)
    ls = n2.encoder.latent_scaling
    assert torch.allclose(ls, torch.tensor([32 / 31 * 2, 32 / 31 * 2]))


def test_forward_requires_hip_device(net):
    xyz, vd = torch.zeros(1, 4, 3), torch.zeros(1, 4, 3)
    with pytest.raises(_lib.PixelNerfHipError):  # CPU module: no fallback, with or without autograd
        net(xyz, coarse=True, viewdirs=vd)
    with torch.no_grad(), pytest.raises(_lib.PixelNerfHipError):
        net(xyz, coarse=True, viewdirs=vd)
    with pytest.raises(_lib.PixelNerfHipError):  # ResnetFC.forward on explicit rows: HIP operators with or without autograd
        net.mlp_coarse(torch.zeros(2, 554))
    with torch.no_grad(), pytest.raises(_lib.PixelNerfHipError):
        net.mlp_coarse(torch.zeros(2, 554))


def test_renderer_construction_and_schedule():
    r = NeRFRenderer.from_conf(default_renderer_conf(), lindisp=False, eval_batch_size=50000)
    assert (r.n_coarse, r.n_fine, r.n_fine_depth, r.depth_std) == (64, 32, 16, 0.01)
    assert r.using_fine and r.sched is None and r.white_bkgd and r.eval_batch_size == 50000
    assert set(r.state_dict()) == {"iter_idx", "last_sched"}
    r2 = NeRFRenderer(n_coarse=8, n_fine=0, sched=[[2, 4], [16, 32], [4, 8]])
    assert not r2.using_fine
    r2.sched_step(3)
    assert (r2.n_coarse, r2.n_fine, int(r2.last_sched)) == (16, 4, 1)
    r2.sched_step(1)
    assert (r2.n_coarse, r2.n_fine, int(r2.last_sched)) == (32, 8, 2)


def test_bind_parallel_and_wrapper(net):
    r = NeRFRenderer(n_coarse=8, n_fine=4)
    w = r.bind_parallel(net, gpus=None, simple_output=True)
    assert isinstance(w, _RenderWrapper) and w.simple_output and w.net is net and w.renderer is r
    assert isinstance(r.bind_parallel(net, [0]), _RenderWrapper)
    rgb, depth = w(torch.zeros(0, 5, 8))  # empty super-batch guard, nerf.py:23-27
    assert rgb.shape == (0, 3) and depth.shape == (0,)
    # >1 GPU without a process group: single-process sharding over the listed devices, like the reference's DataParallel
    assert type(r.bind_parallel(net, [0, 1])).__name__ == "_MultiDeviceRenderWrapper"
    with pytest.raises(AssertionError):
        r(net, torch.zeros(5, 8))  # rays must be (SB,B,8), nerf.py:269


def test_conf_dotmap_and_helpers():
    c = Conf(a=1, sub=dict(b=2.5, l=[1, 2]))
    assert c.get_int("a") == 1 and c.get_int("zz", 7) == 7 and c["sub"].get_float("b") == 2.5
    assert c["sub"].get_list("l") == [1, 2] and c.get_list("sched", None) is None
    d = DotMap(coarse=DotMap(rgb=1))
    assert len(d.fine) == 0 and d.coarse.rgb == 1  # train/train.py:201-202 relies on this
    d.fine = DotMap(rgb=2)
    assert d.toDict() == {"coarse": {"rgb": 1}, "fine": {"rgb": 2}}
    t = torch.arange(6.0).reshape(3, 2)
    assert torch.equal(repeat_interleave(t, 2), t.repeat_interleave(2, 0))
    x = torch.arange(24.0).reshape(6, 4)
    assert torch.equal(combine_interleaved(x, (3, 2)), x.reshape(1, 3, 2, 4).mean(1))
    assert abs(psnr(torch.zeros(4), torch.full((4,), 0.1)) - 20.0) < 1e-4
    rays = gen_rays(torch.eye(4)[None], 4, 3, 2.0, 0.5, 1.5)
    assert rays.shape == (1, 3, 4, 8) and torch.allclose(rays[..., 3:6].norm(dim=-1), torch.ones(1, 3, 4))


def test_pretrained_encoder_without_weights_warns():
    """conf/default.conf asks for pretrained=True; ImageNet weights are not available offline -- that must be loud."""
    import warnings
    from pixelnerf_amd.model.encoder import SpatialEncoder
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        SpatialEncoder(pretrained=True)
    assert any("randomly initialised" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        SpatialEncoder(pretrained=False)
    assert not w


def test_packed_streams_follow_fused_optimizer_updates(net):
    """torch's fused optimizers write parameters in place without bumping tensor._version; the cache key of the packed
    weight streams / folded tables must change anyway (a stale stream would train on frozen weights)."""
    mlp = net.mlp_coarse
    for fused in (False, True):
        opt = torch.optim.Adam(mlp.parameters(), lr=1e-3, fused=fused)
        before = mlp._fingerprint()
        for p in mlp.parameters():
            p.grad = torch.ones_like(p)
        opt.step()
        assert mlp._fingerprint() != before, f"fused={fused}"
        assert mlp._fingerprint() == mlp._fingerprint()


def test_checkpoint_helpers_follow_the_reference_layout(tmp_path):
    """save_weights / load_weights (src/model/models.py:268-316): <checkpoints_path>/<name>/pixel_nerf_latest with one backup
    generation, pixel_nerf_init for fresh runs, --resume picks latest, a missing file warns and leaves the model alone"""
    import types
    import warnings as W
    net = make_model(default_model_conf())
    args = types.SimpleNamespace(checkpoints_path=str(tmp_path), name="exp", resume=True)
    os.makedirs(tmp_path / "exp")
    with W.catch_warnings(record=True) as caught:
        W.simplefilter("always")
        assert net.load_weights(args) is net  # nothing there yet
    assert any("does not exist" in str(w.message) for w in caught)
    w0 = net.mlp_coarse.lin_out.weight.detach().clone()
    net.save_weights(args)
    assert (tmp_path / "exp" / "pixel_nerf_latest").exists() and not (tmp_path / "exp" / "pixel_nerf_backup").exists()
    with torch.no_grad():
        net.mlp_coarse.lin_out.weight.add_(1.0)
    net.save_weights(args)  # the first file becomes the backup
    assert (tmp_path / "exp" / "pixel_nerf_backup").exists()
    other = make_model(default_model_conf())
    other.load_weights(args)
    assert torch.equal(other.mlp_coarse.lin_out.weight, w0 + 1.0)
    backup = torch.load(tmp_path / "exp" / "pixel_nerf_backup", map_location="cpu")
    assert torch.equal(backup["mlp_coarse.lin_out.weight"], w0)
    # init checkpoints: written with opt_init, used by a run that does not resume; opt_init without --resume loads nothing
    net.save_weights(args, opt_init=True)
    assert (tmp_path / "exp" / "pixel_nerf_init").exists()
    fresh = types.SimpleNamespace(checkpoints_path=str(tmp_path), name="exp", resume=False)
    third = make_model(default_model_conf())
    assert third.load_weights(fresh, opt_init=True) is None
    third.load_weights(fresh)
    assert torch.equal(third.mlp_coarse.lin_out.weight, w0 + 1.0)


def test_model_variants_construct_with_the_reference_state_dict_layout():
    """every model conf of testdata.synthetic.VARIANTS (outside the shipped one): d_in / d_latent as the reference derives them
    (models.py:42-58, frozen in tests/golden/variants.npz) and the ResnetFC state_dict layout the generator asserted against the
    reference module; the global encoder is the plain-torch trunk through its average pool"""
    from pixelnerf_amd.model.encoder import ImageEncoder
    from testdata import synthetic
    g = load_golden("variants")
    for name in synthetic.VARIANTS:
        n = make_model(Conf(synthetic.variant_model_conf(name)))
        assert (n.d_in, n.d_latent) == (int(g[f"{name}_d_in"]), int(g[f"{name}_d_latent"])), name
        assert not n.fused_supported()
        for mlp in (n.mlp_coarse, n.mlp_fine):
            shapes = [(k, tuple(v.shape)) for k, v in mlp.state_dict().items()]
            assert shapes == synthetic.resnetfc_shapes(n.d_in, n.d_latent, **synthetic.VARIANTS[name][2]), name
    assert make_model(default_model_conf()).fused_supported()
    enc = ImageEncoder("resnet18", pretrained=False, latent_size=16).eval()
    with torch.no_grad():
        lat = enc(torch.rand(3, 3, 32, 32))
    assert lat.shape == (3, 16) and enc.latent is lat and enc.index(torch.zeros(3, 7, 2)).shape == (3, 16, 7)
    assert "fc.weight" in enc.state_dict() and "model.layer4.1.conv2.weight" in enc.state_dict()
    assert "fc.weight" not in ImageEncoder("resnet18", pretrained=False, latent_size=512).state_dict()


def test_encoder_lookup_modes_gate_the_fused_kernels():
    """the fused lookup is grid_sample(bilinear, border) only: any other `index_interp` / `index_padding` the reference's encoder
    accepts (encoder.py:27-28) must take the composed forward, whose `index` is then ATen's grid_sample"""
    for over in (dict(index_padding="zeros"), dict(index_interp="nearest"), dict(index_padding="reflection")):
        c = default_model_conf()
        c["encoder"] = dict(c["encoder"], **over)
        assert not make_model(Conf(c)).fused_supported(), over
    c = default_model_conf()
    c["encoder"] = dict(c["encoder"], index_padding="zeros")
    enc = make_model(Conf(c)).encoder
    enc.latent = torch.rand(2, 512, 4, 5)
    uv = torch.tensor([[[-1.5, 0.0], [0.3, 0.2], [1.0, 1.0]]]).expand(2, -1, -1)
    got = enc.index(uv)  # a CPU tensor is fine here: this branch is plain torch
    want = torch.nn.functional.grid_sample(enc.latent, uv.unsqueeze(2), align_corners=True, mode="bilinear", padding_mode="zeros")[..., 0]
    assert got.shape == (2, 512, 3) and torch.equal(got, want)


def test_saturation_guard_verdicts_accumulate_until_polled():
    """a guarded call's flag words must survive until a poll reports them -- under PIXELNERF_SATURATION_GUARD=always
    the previous call's copy is still in flight when the next call ends.  The harvest ORs every ARRIVED copy into the carry
    and returns its pinned words to the pool; copies still in flight stay pending (host-side logic, fake events)."""
    from pixelnerf_amd import ops

    class Ev:
        def __init__(self, done):
            self.done, self.waited = done, False

        def query(self):
            return self.done

        def synchronize(self):
            self.done = self.waited = True

    st = dict(pool=[], pending=[(Ev(True), torch.tensor([1 << 2, 0], dtype=torch.int32)),
                                (Ev(False), torch.tensor([1 << 10, 1 << 3], dtype=torch.int32)),
                                (Ev(True), torch.tensor([1 << 4, 1 << 11], dtype=torch.int32))], carry=[0, 0], seen=False)
    ops._sat_harvest(st)
    assert st["carry"] == [(1 << 2) | (1 << 4), 1 << 11] and st["seen"] and len(st["pending"]) == 1 and len(st["pool"]) == 2
    ops._sat_harvest(st)  # the middle copy is still in flight: nothing new
    assert st["carry"] == [(1 << 2) | (1 << 4), 1 << 11] and len(st["pending"]) == 1
    ops._sat_harvest(st, wait=True)
    assert st["carry"] == [(1 << 2) | (1 << 4) | (1 << 10), (1 << 11) | (1 << 3)] and not st["pending"] and len(st["pool"]) == 3


def test_packed_cache_never_blesses_a_stream_it_did_not_fingerprint():
    """ResnetFC._cached, driven with a fake builder on CPU.  A stream packed in the re-pack-every-call phase
    of a training loop carries no content fingerprint.  When such a stream is served a SECOND time under an unchanged cache key
    (a `.data` write moves neither `_version` nor the optimizer-step count) it must be BUILT AGAIN from the live parameters and the
    fingerprint epoch must move (the folded lin_z tables hang on it) -- recording the checksum of the current parameters for the
    old stream would bless a stale stream.  Streams packed under no_grad / in eval mode are fingerprinted at once."""
    mlp = make_model(default_model_conf()).mlp_coarse
    built, recorded, verified = [], [], []
    mlp._content_record = lambda key: recorded.append(key)
    mlp._content_verify = lambda key: verified.append(key) or False

    def build(out):
        built.append(out)
        return object()

    key = ("f16x3", True)
    mlp.train()
    with torch.enable_grad():
        s1 = mlp._cached(key, "f16x3", build)                       # first pack: fingerprinted
        assert len(built) == 1 and recorded == [key]
        mlp.__dict__["_opt_steps"] = mlp.__dict__.get("_opt_steps", 0) + 1   # an optimizer step
        s2 = mlp._cached(key, "f16x3", build)                       # replaces a never re-used stream: no fingerprint
        assert s2 is not s1 and len(built) == 2 and recorded == [key]
        epoch = mlp.__dict__.get("_epoch", 0)
        s3 = mlp._cached(key, "f16x3", build)                       # RE-USE of the unfingerprinted stream: built again, not blessed
        assert s3 is not s2 and len(built) == 3 and recorded == [key, key]
        assert mlp.__dict__["_epoch"] == epoch + 1                  # dependants (folded tables) re-derive from the live parameters
        assert verified == []
        s4 = mlp._cached(key, "f16x3", build)                       # now a fingerprinted stream: served from the cache, verified
        assert s4 is s3 and len(built) == 3 and verified == [key]
        mlp.__dict__["_opt_steps"] += 1
        mlp._cached(key, "f16x3", build)                            # replaces a stream that WAS re-used: fingerprinted from the start
        assert len(built) == 4 and recorded == [key, key, key]
    mlp.__dict__["_opt_steps"] += 1
    mlp._cached(key, "f16x3", build)
    mlp.__dict__["_opt_steps"] += 1
    with torch.no_grad():                                           # validation pass between steps: always fingerprinted
        n = len(recorded)
        mlp._cached(key, "f16x3", build)
        assert len(recorded) == n + 1
    # the differentiable path packs INSIDE torch.autograd.Function.forward / backward, where grad mode is off: it says what it is
    # (training_pass=True) instead of being taken for a validation pass -- a training loop must not fingerprint every step
    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            assert not torch.is_grad_enabled()
            mlp.__dict__["_opt_steps"] += 1
            mlp._cached(key, "f16x3", build, True)
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            return g * 2.0

    mlp.__dict__["_opt_steps"] += 1
    mlp._cached(key, "f16x3", build, True)                          # (replaces the validation pass's stream)
    for _ in range(3):
        n, b = len(recorded), len(built)
        _Fn.apply(torch.ones(1, requires_grad=True)).backward()
        assert len(recorded) == n and len(built) == b + 1           # re-packed every step, never fingerprinted
    mlp.eval()
    mlp.__dict__["_opt_steps"] += 1
    n = len(recorded)
    mlp._cached(key, "f16x3", build)                                # eval mode: likewise
    assert len(recorded) == n + 1
    # the exact-fp32 form holds pointers, not a copy: never fingerprinted, never re-built on a hit
    b0 = len(built)
    a = mlp._cached("f32-key", "f32", build)
    assert mlp._cached("f32-key", "f32", build) is a and len(built) == b0 + 1
