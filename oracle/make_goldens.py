#!/usr/bin/env python3
"""
Golden-vector generator (TEST INFRASTRUCTURE; runs only in the build container).

Imports the UNMODIFIED reference implementation from /root/reference/src (through the
import stubs in oracle/ref_stubs/ for pyhocon / cv2 / torchvision / dotmap, none of which is
used by the hot path at run time -- SURVEY.md Appendix B), builds the reference
`PixelNeRFNet` + `NeRFRenderer`, loads the seeded synthetic weights / scenes of
pixelnerf_amd.synthetic, pins every random draw by substituting pre-drawn noise in the
reference's draw order, runs `NeRFRenderer.forward` on CPU and freezes inputs + outputs
(+ intermediate z samples and per-point rgb/sigma captured by wrapping `composite`) into
tests/golden/<scenario>.npz.

The reference repository has no tests or golden vectors of its own, so these
fixtures ARE the parity pin for oracle/pnr_oracle.py and, through it, for the HIP path.

Usage:  python oracle/make_goldens.py            (writes tests/golden/*.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PIXELNERF_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
sys.path.insert(1, os.path.join(REF, "src"))
sys.path.insert(2, ROOT)

from testdata import synthetic  # noqa: E402

# name: (scene, n_coarse, n_fine, n_fine_depth, rays/object, lindisp, use mlp_fine)
SCENARIOS = {
    "sn64_c32":        ("sn64", 32, 0, 0, 192, False, True),     # BASELINE config (1) sampling
    "sn64_64_128":     ("sn64", 64, 128, 16, 96, False, True),   # BASELINE metric sampling
    "srn_mini_64_128": ("srn_mini", 64, 128, 16, 64, False, True),   # NS=2 mean pooling
    "dtu_mini_64_128": ("dtu_mini", 64, 128, 16, 64, False, True),   # NS=3, black bkgd, fx!=fy
    "train_64_32":     ("train", 64, 32, 16, 32, False, True),   # SB=4 (config 5 shapes)
    "mv_mini_lindisp": ("mv_mini", 32, 16, 0, 32, True, True),   # SB=2 x NS=2, lindisp, Kfd=0
    "sn64_coarse_only_mlp": ("sn64", 16, 16, 16, 32, False, False),  # mlp_fine=None, Kf-Kfd=0
    "dtu6_mini_64_128": ("dtu6_mini", 64, 128, 16, 48, False, True),   # NS=6: the reference's 6-view DTU eval (README.md:201)
    "dtu9_mini_64_128": ("dtu9_mini", 64, 128, 16, 48, False, True),   # NS=9: the 9-view eval (README.md:202)
}
# adversarial scenarios: surface-like density (synthetic.surface_variant: sigma ~0 off a thin shell,
# 50..300 on it -> peaked coarse weights), importance draw u2 = 1 - 2^-24 on column 0 of every 4th ray (searchsorted past the last cdf
# entry whenever cdf[-1] rounds below 1: ind == Kc, a fine sample beyond `far`, negative last delta, nerf.py:138-141,181)
ADVERSARIAL = {
    "adv_surface_sn64":   ("sn64", 64, 128, 16, 96, False, True, dict(gain=100.0, tau=4.3)),
    "adv_surface_srn":    ("srn_mini", 64, 128, 16, 64, False, True, dict(gain=100.0, tau=3.9)),
    "adv_surface_dtu":    ("dtu_mini", 64, 128, 16, 64, False, True, dict(gain=100.0, tau=3.9)),
    "adv_surface_coarse_net": ("sn64", 64, 128, 16, 64, False, False, dict(gain=100.0, tau=4.3)),  # mlp_fine=None re-use path
}
SCENARIOS.update({k: v[:7] for k, v in ADVERSARIAL.items()})
MLP_SEED_COARSE, MLP_SEED_FINE, SCENE_SEED = 11, 12, 2


class Conf(dict):
    """dict-backed stand-in for a pyhocon ConfigTree (get_* with defaults, conf['sub'])."""

    def _get(self, k, d=None):
        return self[k] if k in self else d

    get_bool = get_int = get_float = get_string = get_list = _get

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return Conf(v) if isinstance(v, dict) and not isinstance(v, Conf) else v


def model_conf():
    """conf/default.conf:3-48 + conf/default_mv.conf:3-22 (the one shape every shipped
    experiment resolves to)."""
    mlp = dict(type="resnet", n_blocks=5, d_hidden=512, combine_layer=3, combine_type="average")
    return Conf(
        use_encoder=True, use_global_encoder=False, use_xyz=True, canon_xyz=False,
        use_code=True, code=dict(num_freqs=6, freq_factor=1.5, include_input=True),
        use_viewdirs=True, use_code_viewdirs=False,
        mlp_coarse=dict(mlp), mlp_fine=dict(mlp),
        encoder=dict(backbone="resnet34", pretrained=False, num_layers=4),
    )


class _TorchProxy:
    """Stands in for the `torch` global of reference module render.nerf: forwards everything
    to torch but serves rand / rand_like / randn_like from a pre-drawn queue."""

    def __init__(self, queue):
        self._q = queue

    def __getattr__(self, k):
        return getattr(torch, k)

    def _pop(self, kind, shape):
        k, t = self._q.pop(0)
        assert k == kind, (k, kind)
        assert tuple(t.shape) == tuple(shape), (k, t.shape, shape)
        return t.clone()

    def rand(self, *shape, **kw):
        return self._pop("rand", shape)

    def rand_like(self, t):
        return self._pop("rand_like", t.shape)

    def randn_like(self, t):
        return self._pop("randn_like", t.shape)


def build_reference_net(use_fine, surface=None):
    import model as ref_model

    def params(seed):
        p = synthetic.make_mlp_params(seed)
        return p if surface is None else synthetic.surface_variant(p, surface["gain"], surface["tau"])

    net = ref_model.make_model(model_conf())
    net.mlp_coarse.load_state_dict(params(MLP_SEED_COARSE))
    if use_fine:
        net.mlp_fine.load_state_dict(params(MLP_SEED_FINE))
    else:
        net.mlp_fine = None  # eval/eval.py:140
    return net.eval()


def set_encode_state(net, scene):
    """What PixelNeRFNet.encode() leaves behind (models.py:111-141, encoder.py:160-163)."""
    lat = scene["latent"]
    net.encoder.latent = lat
    ls = torch.tensor([lat.shape[-1], lat.shape[-2]], dtype=torch.float32)
    net.encoder.latent_scaling = ls / (ls - 1) * 2.0
    net.poses = scene["poses"]
    net.image_shape = scene["image_shape"]
    net.focal = scene["focal"]
    net.c = scene["c"]
    net.num_objs = scene["SB"]
    net.num_views_per_obj = scene["NS"]


def run_scenario(name):
    import render.nerf as ref_nerf

    scene_name, Kc, Kf, Kfd, n_rays, lindisp, use_fine = SCENARIOS[name]
    scene, meta = synthetic.make_scene(scene_name, seed=SCENE_SEED)
    rays = synthetic.target_rays(meta, n_rays=n_rays)  # (SB, n_rays, 8)
    SB = rays.shape[0]
    R = SB * n_rays
    noise = synthetic.make_noise(R, Kc, Kf, Kfd)
    surface = ADVERSARIAL[name][7] if name in ADVERSARIAL else None
    if surface is not None and "u2" in noise:
        noise["u2"][::4, 0] = float(np.float32(1.0) - np.float32(2.0 ** -24))  # every 4th ray: the largest uniform draw

    queue = [("rand_like", noise["u1"])]
    if Kf > 0:
        if Kf - Kfd > 0:
            queue += [("rand", noise["u2"]), ("rand_like", noise["u3"])]
        if Kfd > 0:
            queue += [("randn_like", noise["n4"])]

    net = build_reference_net(use_fine, surface)
    set_encode_state(net, scene)
    renderer = ref_nerf.NeRFRenderer(
        n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=0.01,
        white_bkgd=meta["white_bkgd"], lindisp=lindisp, eval_batch_size=50000,
    ).eval()

    captured = []
    orig_composite = renderer.composite

    def composite_spy(model, rays_, z_samp, coarse=True, sb=0):
        outs = []
        orig_forward = model.forward

        def fwd_spy(*a, **k):
            o = orig_forward(*a, **k)
            outs.append(o)
            return o

        model.forward = fwd_spy
        try:
            res = orig_composite(model, rays_, z_samp, coarse=coarse, sb=sb)
        finally:
            del model.forward
        out = torch.cat(outs, dim=1).reshape(z_samp.shape[0], z_samp.shape[1], 4)
        captured.append((z_samp.clone(), out.clone()))
        return res

    renderer.composite = composite_spy
    real_torch = ref_nerf.torch
    ref_nerf.torch = _TorchProxy(queue)
    try:
        with torch.no_grad():
            out = renderer(net, rays, want_weights=True)
    finally:
        ref_nerf.torch = real_torch
    assert len(queue) == 0, "noise queue not fully consumed"

    rec = dict(
        scene=scene_name, n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, lindisp=int(lindisp),
        use_mlp_fine=int(use_fine), white_bkgd=int(meta["white_bkgd"]), depth_std=0.01,
        mlp_seed_coarse=MLP_SEED_COARSE, mlp_seed_fine=MLP_SEED_FINE, scene_seed=SCENE_SEED,
        rays=rays.numpy(),
    )
    if surface is not None:
        rec["sigma_gain"], rec["sigma_tau"] = surface["gain"], surface["tau"]
    for k, v in noise.items():
        rec["noise_" + k] = v.numpy()
    rec["coarse_rgb"] = out.coarse.rgb.numpy()
    rec["coarse_depth"] = out.coarse.depth.numpy()
    rec["coarse_weights"] = out.coarse.weights.numpy()
    rec["coarse_z"] = captured[0][0].numpy()
    rec["coarse_rgbsigma"] = captured[0][1].numpy()
    if Kf > 0:
        rec["fine_rgb"] = out.fine.rgb.numpy()
        rec["fine_depth"] = out.fine.depth.numpy()
        rec["fine_weights"] = out.fine.weights.numpy()
        rec["fine_z"] = captured[1][0].numpy()
        rec["fine_rgbsigma"] = captured[1][1].numpy()
    return rec


def stage_goldens():
    """Stage-level fixtures from the reference modules themselves: PositionalEncoding
    (src/model/code.py), SpatialEncoder.index (src/model/encoder.py:80-109) and
    PixelNeRFNet.forward (src/model/models.py:146-266) on seeded random query points."""
    from model.code import PositionalEncoding

    rs = np.random.RandomState(99)
    rec = {}
    x = torch.from_numpy(rs.uniform(-3, 3, (257, 3)).astype(np.float32))
    code = PositionalEncoding(num_freqs=6, d_in=3, freq_factor=1.5, include_input=True)
    rec["posenc_x"] = x.numpy()
    rec["posenc_out"] = code(x).numpy()

    for scene_name in ("sn64", "dtu_mini", "mv_mini"):
        scene, meta = synthetic.make_scene(scene_name, seed=SCENE_SEED)
        net = build_reference_net(True)
        set_encode_state(net, scene)
        SB, NS = scene["SB"], scene["NS"]
        B = 200
        # query points around the object, plus some that project outside the source images
        xyz = torch.from_numpy(rs.uniform(-1.0, 1.0, (SB, B, 3)).astype(np.float32))
        xyz[:, :20] *= 4.0
        vd = torch.from_numpy(rs.randn(SB, B, 3).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        with torch.no_grad():
            out_c = net(xyz, coarse=True, viewdirs=vd)
            out_f = net(xyz, coarse=False, viewdirs=vd)
            uv = torch.from_numpy(
                rs.uniform(-8, meta["W"] + 8, (SB * NS, 64, 2)).astype(np.float32))
            idx = net.encoder.index(uv, None, net.image_shape)
        rec[f"{scene_name}_xyz"] = xyz.numpy()
        rec[f"{scene_name}_viewdirs"] = vd.numpy()
        rec[f"{scene_name}_out_coarse"] = out_c.numpy()
        rec[f"{scene_name}_out_fine"] = out_f.numpy()
        rec[f"{scene_name}_uv"] = uv.numpy()
        rec[f"{scene_name}_index"] = idx.numpy()
    return rec


def manyview_stage_goldens():
    """stage_goldens' PixelNeRFNet.forward / SpatialEncoder.index fixtures on the 6- and 9-view scenes (models.py:102-105
    num_views_per_obj = 6 / 9; resnetfc.py:168-172 the mean over 6 / 9 rows), plus the view MAXIMUM over 9 rows (util.py:467-468).
    A file of its own (stages_manyview.npz) so that stages.npz keeps regenerating bit-identically."""
    rs = np.random.RandomState(199)
    rec = {}
    for scene_name in ("dtu6_mini", "dtu9_mini"):
        scene, meta = synthetic.make_scene(scene_name, seed=SCENE_SEED)
        net = build_reference_net(True)
        set_encode_state(net, scene)
        SB, NS = scene["SB"], scene["NS"]
        B = 200
        xyz = torch.from_numpy(rs.uniform(-1.0, 1.0, (SB, B, 3)).astype(np.float32))
        xyz[:, :20] *= 4.0
        vd = torch.from_numpy(rs.randn(SB, B, 3).astype(np.float32))
        vd = vd / vd.norm(dim=-1, keepdim=True)
        with torch.no_grad():
            out_c = net(xyz, coarse=True, viewdirs=vd)
            out_f = net(xyz, coarse=False, viewdirs=vd)
            uv = torch.from_numpy(rs.uniform(-8, meta["W"] + 8, (SB * NS, 64, 2)).astype(np.float32))
            idx = net.encoder.index(uv, None, net.image_shape)
            net.mlp_coarse.combine_type = net.mlp_fine.combine_type = "max"
            max_c = net(xyz, coarse=True, viewdirs=vd)
            max_f = net(xyz, coarse=False, viewdirs=vd)
        rec[f"{scene_name}_xyz"] = xyz.numpy()
        rec[f"{scene_name}_viewdirs"] = vd.numpy()
        rec[f"{scene_name}_out_coarse"] = out_c.numpy()
        rec[f"{scene_name}_out_fine"] = out_f.numpy()
        rec[f"{scene_name}_max_coarse"] = max_c.numpy()
        rec[f"{scene_name}_max_fine"] = max_f.numpy()
        rec[f"{scene_name}_uv"] = uv.numpy()
        rec[f"{scene_name}_index"] = idx.numpy()
    return rec


def combine_max_goldens():
    """PixelNeRFNet.forward of the UNMODIFIED reference with `combine_type = "max"` on both ResnetFCs (util.combine_interleaved's
    other branch, src/util/util.py:467-468; resnetfc.py:168-172) on the multi-view scenes' stage points (tests/golden/stages.npz
    holds the points): the pin of the oracle's and the kernels' view maximum."""
    st = np.load(os.path.join(ROOT, "tests", "golden", "stages.npz"))
    rec = {}
    for scene_name in ("dtu_mini", "mv_mini"):
        scene, meta = synthetic.make_scene(scene_name, seed=SCENE_SEED)
        net = build_reference_net(True)
        net.mlp_coarse.combine_type = net.mlp_fine.combine_type = "max"
        set_encode_state(net, scene)
        xyz, vd = torch.from_numpy(st[f"{scene_name}_xyz"]), torch.from_numpy(st[f"{scene_name}_viewdirs"])
        with torch.no_grad():
            rec[f"{scene_name}_out_coarse"] = net(xyz, coarse=True, viewdirs=vd).numpy()
            rec[f"{scene_name}_out_fine"] = net(xyz, coarse=False, viewdirs=vd).numpy()
    return rec


def variants_goldens():
    """PixelNeRFNet.forward of the UNMODIFIED reference under model confs OUTSIDE the shipped one (testdata.synthetic.VARIANTS:
    use_code_viewdirs=True -- the reference's default --, normalize_z=False, use_xyz=False, Softplus + SPADE + max pooling, a global
    encoder, no encoder at all, other ResnetFC shapes): the pin of the oracle's general forward and of the composed HIP path.
    Inputs and weights are seeded functions (synthetic.variant_inputs / fill_state) -- the fixture holds the outputs only."""
    import model as ref_model
    rec = {}
    for name in synthetic.VARIANTS:
        scene, meta, xyz, vd, glob = synthetic.variant_inputs(name)
        net = ref_model.make_model(Conf(synthetic.variant_model_conf(name))).eval()
        for i, mlp in enumerate((net.mlp_coarse, net.mlp_fine)):
            shapes = [(k, tuple(v.shape)) for k, v in mlp.state_dict().items()]
            assert shapes == synthetic.resnetfc_shapes(net.d_in, net.d_latent, **synthetic.VARIANTS[name][2]), name
            mlp.load_state_dict(synthetic.fill_state(shapes, synthetic.VARIANT_SEED + i))
        set_encode_state(net, scene)
        if glob is not None:
            net.global_encoder.latent = glob  # what ImageEncoder.forward leaves behind (encoder.py:220)
        with torch.no_grad():
            rec[f"{name}_out_coarse"] = net(xyz, coarse=True, viewdirs=vd).numpy()
            rec[f"{name}_out_fine"] = net(xyz, coarse=False, viewdirs=vd).numpy()
        rec[f"{name}_d_in"] = np.int64(net.d_in)
        rec[f"{name}_d_latent"] = np.int64(net.d_latent)
    return rec


def plane_goldens():
    """Points ON and BEHIND a source camera's image plane (SURVEY App. A "known sharp edges": no frustum culling,
    `xc.z == 0` divides by zero, `xc.z > 0` mirrors; models.py:206-212,237-239) through the reference's
    PixelNeRFNet.forward.  Scene plane_mini: source camera 0 is axis-aligned (world y = 2 is its plane z_cam = 0), so
    the degenerate coordinates are exact in fp32: u = -x/0 = +-inf (border-clamped by grid_sample), 0/0 = NaN."""
    rs = np.random.RandomState(5)
    scene, meta = synthetic.make_scene("plane_mini", seed=SCENE_SEED)
    net = build_reference_net(True)
    set_encode_state(net, scene)
    B = 64
    xyz = rs.uniform(-1.0, 1.0, (1, B, 3)).astype(np.float32)
    xyz[0, :16, 1] = 2.0                         # on camera 0's plane: uv = (+-inf, +-inf)
    xyz[0, 0] = (0.0, 2.0, 0.0)                  # camera 0's centre: 0/0 in both coordinates
    xyz[0, 1] = (0.3, 2.0, 0.0)                  # u = +inf, v = 0/0
    xyz[0, 2] = (0.0, 2.0, -0.4)                 # u = 0/0, v = +-inf
    xyz[0, 16:32, 1] = 2.0 + rs.uniform(0.01, 1.5, 16).astype(np.float32)  # behind camera 0 (mirrored projection)
    xyz = torch.from_numpy(xyz)
    vd = torch.from_numpy(rs.randn(1, B, 3).astype(np.float32))
    vd = vd / vd.norm(dim=-1, keepdim=True)
    with torch.no_grad():
        out_c = net(xyz, coarse=True, viewdirs=vd)
        out_f = net(xyz, coarse=False, viewdirs=vd)
    return dict(scene="plane_mini", xyz=xyz.numpy(), viewdirs=vd.numpy(), out_coarse=out_c.numpy(), out_fine=out_f.numpy())


class _Const(torch.nn.Module):
    """Trunk stand-in: returns a fixed tensor whatever it is fed (the torchvision backbone is a stub here)."""

    def __init__(self, t=None):
        super().__init__()
        self.t = t

    def forward(self, x):
        return x if self.t is None else self.t


def neighbour_goldens():
    """Fixtures for the rows next to the hot path, from the reference's own code:
    * SpatialEncoder.forward's output formatting (src/model/encoder.py:150-163) run on seeded ResNet-stage
      tensors (the trunk modules are replaced by constants; only interpolate + cat + latent_scaling run);
    * util.gen_rays (src/util/util.py:238-276) with fx != fy and an off-centre principal point;
    * util.psnr (src/util/util.py:474-481)."""
    import util as ref_util
    from model.encoder import SpatialEncoder

    rec = {}
    for name in ("pool", "nopool"):  # the full-size "dtu" pyramid is a property/bench case, not a fixture
        NV, shapes = synthetic.PYRAMIDS[name]
        stages = synthetic.pyramid_stages(name)
        enc = SpatialEncoder(backbone="resnet34", pretrained=False, num_layers=4, use_first_pool=(name == "pool"))
        m = enc.model
        m.conv1, m.bn1, m.relu = _Const(stages[0]), _Const(), _Const()
        m.maxpool = _Const()
        m.layer1, m.layer2, m.layer3 = _Const(stages[1]), _Const(stages[2]), _Const(stages[3])
        with torch.no_grad():
            lat = enc(torch.zeros(NV, 3, 8, 8))
        rec[f"pyr_{name}_latent"] = lat.numpy()
        rec[f"pyr_{name}_scaling"] = enc.latent_scaling.numpy()

    rs = np.random.RandomState(77)
    poses = torch.stack([torch.as_tensor(synthetic.pose_spherical(t, p, 2.2)) for t, p in ((10.0, -20.0), (130.0, -35.0), (250.0, 5.0))])
    focal = torch.tensor([41.5, 39.25])
    c = torch.tensor([10.75, 6.5])
    rec["rays_poses"] = poses.numpy()
    rec["rays_focal"], rec["rays_c"] = focal.numpy(), c.numpy()
    rec["rays_out"] = ref_util.gen_rays(poses, 20, 15, focal, 0.8, 1.8, c=c).numpy()

    # util.bbox_sample draws randint, rand, rand in this order: replay them as explicit inputs
    bboxes = torch.tensor([[3.0, 2.0, 17.0, 12.0], [0.0, 0.0, 19.0, 14.0], [8.0, 5.0, 8.0, 5.0]])
    torch.manual_seed(31)
    rec["bbox_pix"] = ref_util.bbox_sample(bboxes, 500).numpy()
    torch.manual_seed(31)
    rec["bbox_ids"] = torch.randint(0, 3, (500,)).numpy()
    rec["bbox_ux"], rec["bbox_uy"] = torch.rand(500).numpy(), torch.rand(500).numpy()
    rec["bbox_boxes"] = bboxes.numpy()

    pred = torch.from_numpy(rs.uniform(-0.2, 1.2, (3, 300, 3)).astype(np.float32))
    gt = torch.from_numpy(rs.uniform(0.0, 1.0, (3, 300, 3)).astype(np.float32))
    rec["psnr_pred"], rec["psnr_gt"] = pred.numpy(), gt.numpy()
    rec["psnr_out"] = np.array([ref_util.psnr(pred[i].clamp(0, 1), gt[i]) for i in range(3)], np.float64)
    return rec


GRAD_SCENARIOS = ("train_64_32", "srn_mini_64_128", "train_cfg5")  # SB=4 x NS=1 (config 5 shapes), NS=2 pooling, config 5 at FULL size
# 3 source views -- what the reference trains DTU with (README.md:204; train/train.py:28,77,138-160 --nviews): a file of its own
# (gradients_3view.npz) so that gradients.npz keeps regenerating bit-identically
GRAD_SCENARIOS_3VIEW = ("dtu_mini_64_128", "train_mv3")  # SB=1 x NS=3; SB=2 x NS=3 (object-major rows under the 3-row mean)
# gradient-only scenarios (no render fixture: rays and noise are regenerated from their seeds by the tests, exactly as below)
GRAD_ONLY = {"train_cfg5": ("train", 64, 32, 16, 128, False, True),  # BASELINE configs[4]: 4 objects x 128 rays, 64+32(16)
             "train_mv3": ("train_mv3", 64, 32, 16, 64, False, True)}  # DTU-style step: 2 objects x 3 views x 64 rays, 64+32(16)

def gradient_goldens(names=GRAD_SCENARIOS):
    """Gradients of the UNMODIFIED reference (torch autograd through NeRFRenderer.forward, train/train.py:199-215:
    MSE(coarse rgb) + MSE(fine rgb) against a seeded target) w.r.t. every ResnetFC parameter of both networks and
    encoder.latent, on the golden scenarios' rays and noise.  Frozen per tensor: L2 norm + a seeded subsample."""
    import render.nerf as ref_nerf

    rec = {}
    for name in names:
        scene_name, Kc, Kf, Kfd, n_rays, lindisp, use_fine = SCENARIOS[name] if name in SCENARIOS else GRAD_ONLY[name]
        scene, meta = synthetic.make_scene(scene_name, seed=SCENE_SEED)
        rays = synthetic.target_rays(meta, n_rays=n_rays)
        SB = rays.shape[0]
        noise = synthetic.make_noise(SB * n_rays, Kc, Kf, Kfd)
        queue = [("rand_like", noise["u1"]), ("rand", noise["u2"]), ("rand_like", noise["u3"]), ("randn_like", noise["n4"])]
        net = build_reference_net(use_fine).train()
        scene = dict(scene)
        scene["latent"] = scene["latent"].clone().requires_grad_(True)
        set_encode_state(net, scene)
        renderer = ref_nerf.NeRFRenderer(n_coarse=Kc, n_fine=Kf, n_fine_depth=Kfd, depth_std=0.01,
                                         white_bkgd=meta["white_bkgd"], lindisp=lindisp, eval_batch_size=50000).train()
        gt = torch.from_numpy(np.random.RandomState(77).uniform(0, 1, (SB, n_rays, 3)).astype(np.float32))
        real_torch = ref_nerf.torch
        ref_nerf.torch = _TorchProxy(queue)
        try:
            out = renderer(net, rays, want_weights=True)
        finally:
            ref_nerf.torch = real_torch
        assert len(queue) == 0
        loss = ((out.coarse.rgb - gt) ** 2).mean() + ((out.fine.rgb - gt) ** 2).mean()
        loss.backward()
        rec[f"{name}_loss"] = np.float64(loss.item())
        rec[f"{name}_gt"] = gt.numpy()
        tensors = [("latent", scene["latent"].grad)]
        for which, mlp in (("coarse", net.mlp_coarse), ("fine", net.mlp_fine)):
            tensors += [(f"{which}.{k}", p.grad) for k, p in mlp.named_parameters()]
        for key, g in tensors:
            flat = g.detach().reshape(-1).numpy()
            idx = synthetic.grad_sample_index(flat.size, key)
            rec[f"{name}_grad_{key}_norm"] = np.float64(np.linalg.norm(flat.astype(np.float64)))
            rec[f"{name}_grad_{key}_sample"] = flat[idx].astype(np.float32)
    return rec


def state_dict_manifest():
    """Names and shapes of the reference net's state_dict (encoder.* excluded: the torchvision
    backbone is a stub here) and of the renderer's -- the checkpoint-compatibility contract
    (SURVEY.md §5 'Checkpoint / resume')."""
    import render.nerf as ref_nerf

    net = build_reference_net(True)
    lines = []
    for k, v in net.state_dict().items():
        if not k.startswith("encoder."):
            lines.append("net %s %s" % (k, "x".join(str(d) for d in v.shape)))
    for k, v in ref_nerf.NeRFRenderer(n_coarse=64, n_fine=32).state_dict().items():
        lines.append("renderer %s %s" % (k, "x".join(str(d) for d in v.shape) or "scalar"))
    return lines


def main():
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count() or 1)
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    names = sys.argv[1:] or (list(SCENARIOS) + ["stages", "stages_manyview", "adv_plane", "neighbours", "gradients", "gradients_3view", "combine_max",
                                "variants", "manifest"])
    for name in names:
        if name == "manifest":
            path = os.path.join(outdir, "state_dict_manifest.txt")
            open(path, "w").write("\n".join(state_dict_manifest()) + "\n")
            print("wrote", path)
            continue
        rec = (stage_goldens() if name == "stages" else neighbour_goldens() if name == "neighbours" else plane_goldens() if name == "adv_plane"
               else manyview_stage_goldens() if name == "stages_manyview" else gradient_goldens(GRAD_SCENARIOS_3VIEW) if name == "gradients_3view"
               else gradient_goldens() if name == "gradients" else combine_max_goldens() if name == "combine_max"
               else variants_goldens() if name == "variants" else run_scenario(name))
        path = os.path.join(outdir, name + ".npz")
        np.savez_compressed(path, **rec)
        print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
