"""
Seeded synthetic scenes (SURVEY.md §8d "common synthetic scene recipe").

There are no datasets or checkpoints in this environment, so every parity test, golden
fixture and benchmark uses these generators.  Everything is produced with
numpy.random.RandomState (bit-stable across numpy versions and machines), so the GPU box
regenerates exactly the tensors the golden fixtures were made from; only rays / noise /
outputs are stored in tests/golden/.

Camera conventions follow the reference: poses are camera-to-world 4x4
(src/util/util.py:309-323 pose_spherical), rays are [origin(3), dir(3), near, far]
(src/util/util.py:238-276 gen_rays).
"""
import math

import numpy as np
import torch

MLP_SHAPE = dict(d_in=42, d_latent=512, d_hidden=512, n_blocks=5, combine_layer=3, d_out=4)


def mlp_param_names(n_blocks=5, combine_layer=3):
    """State-dict key order of the reference ResnetFC (src/model/resnetfc.py:66-130)."""
    names = ["lin_in", "lin_out"]
    for b in range(n_blocks):
        names += [f"blocks.{b}.fc_0", f"blocks.{b}.fc_1"]
    for b in range(min(combine_layer, n_blocks)):
        names.append(f"lin_z.{b}")
    return names


def make_mlp_params(seed, d_in=42, d_latent=512, d_hidden=512, n_blocks=5, combine_layer=3,
                    d_out=4):
    """Random ResnetFC parameters as {state_dict_key: float32 torch tensor}.

    Kaiming-normal fan-in scale for every Linear (reference init, resnetfc.py:36-39,89-94,
    116-117) except fc_1, which the reference zero-initialises; here fc_1 ~ N(0, 0.03^2)
    and all biases ~ N(0, 0.01^2) so that every term of the network is exercised
    (SURVEY.md §8a R10)."""
    rs = np.random.RandomState(seed)
    p = {}

    def lin(name, fan_out, fan_in, std=None):
        s = math.sqrt(2.0 / fan_in) if std is None else std
        p[name + ".weight"] = torch.from_numpy((rs.randn(fan_out, fan_in) * s).astype(np.float32))
        p[name + ".bias"] = torch.from_numpy((rs.randn(fan_out) * 0.01).astype(np.float32))

    lin("lin_in", d_hidden, d_in)
    lin("lin_out", d_out, d_hidden)
    for b in range(n_blocks):
        lin(f"blocks.{b}.fc_0", d_hidden, d_hidden)
        lin(f"blocks.{b}.fc_1", d_hidden, d_hidden, std=0.03)
    for b in range(min(combine_layer, n_blocks)):
        lin(f"lin_z.{b}", d_hidden, d_latent)
    return p


def surface_variant(params, gain, tau):
    """Surface-like density from a random network (adversarial fixtures): only the sigma row of lin_out is changed,
    sigma' = relu(gain * (s - tau)) with s the network's own pre-activation density.  With tau near the 90th
    percentile of s and gain ~100 the density is 0 on ~90 % of the samples and 50..300 on a thin shell, so the coarse
    weights are PEAKED (a few bins carry everything), the inverse-CDF has long flat stretches and steep steps, and a
    16-bit error in s is amplified gain-fold right at the shell's edge -- the regime a trained model renders in."""
    q = {k: v.clone() for k, v in params.items()}
    q["lin_out.weight"][3] *= gain
    q["lin_out.bias"][3] = gain * (q["lin_out.bias"][3] - tau)
    return q


# ---------------------------------------------------------------------------- cameras


def pose_spherical(theta, phi, radius):
    """Camera-to-world pose on a sphere; restates src/util/util.py:279-323."""
    def trans_t(t):
        return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, t], [0, 0, 0, 1]], np.float32)

    def rot_phi(a):
        return np.array([[1, 0, 0, 0], [0, np.cos(a), -np.sin(a), 0],
                         [0, np.sin(a), np.cos(a), 0], [0, 0, 0, 1]], np.float32)

    def rot_theta(a):
        return np.array([[np.cos(a), 0, -np.sin(a), 0], [0, 1, 0, 0],
                         [np.sin(a), 0, np.cos(a), 0], [0, 0, 0, 1]], np.float32)

    c2w = trans_t(radius)
    c2w = rot_phi(phi / 180.0 * np.pi) @ c2w
    c2w = rot_theta(theta / 180.0 * np.pi) @ c2w
    c2w = np.array([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32) @ c2w
    return torch.from_numpy(c2w.astype(np.float32))


def coord_from_blender():
    """src/util/util.py:146-157."""
    return torch.tensor([[1, 0, 0, 0], [0, 0, 1, 0], [0, -1, 0, 0], [0, 0, 0, 1]],
                        dtype=torch.float32)


def unproj_map(width, height, f, c=None):
    """src/util/util.py:113-143: unit camera-space ray per pixel, (H, W, 3)."""
    if c is None:
        c = [width * 0.5, height * 0.5]
    fx, fy = (f, f) if isinstance(f, (float, int)) else (float(f[0]), float(f[1]))
    Y, X = torch.meshgrid(
        torch.arange(height, dtype=torch.float32) - float(c[1]),
        torch.arange(width, dtype=torch.float32) - float(c[0]),
        indexing="ij",
    )
    X = X / float(fx)
    Y = Y / float(fy)
    Z = torch.ones_like(X)
    unproj = torch.stack((X, -Y, -Z), dim=-1)
    unproj /= torch.norm(unproj, dim=-1).unsqueeze(-1)
    return unproj


def gen_rays(poses, width, height, focal, z_near, z_far, c=None):
    """src/util/util.py:238-276 (ndc=False): (B, H, W, 8) rays."""
    num_images = poses.shape[0]
    cam_unproj_map = unproj_map(width, height, focal, c=c).unsqueeze(0).repeat(num_images, 1, 1, 1)
    cam_centers = poses[:, None, None, :3, 3].expand(-1, height, width, -1)
    cam_raydir = torch.matmul(poses[:, None, None, :3, :3], cam_unproj_map.unsqueeze(-1))[..., 0]
    cam_nears = torch.full((num_images, height, width, 1), float(z_near))
    cam_fars = torch.full((num_images, height, width, 1), float(z_far))
    return torch.cat((cam_centers, cam_raydir, cam_nears, cam_fars), dim=-1)


def encode_state(src_poses_c2w, focal, c, W, H):
    """The camera part of PixelNeRFNet.encode (src/model/models.py:112-141): world->camera
    [R^T | -R^T t], focal (fx, -fy), principal point, image_shape (W, H)."""
    rot = src_poses_c2w[:, :3, :3].transpose(1, 2)
    trans = -torch.bmm(rot, src_poses_c2w[:, :3, 3:])
    poses = torch.cat((rot, trans), dim=-1).contiguous()
    focal_t = torch.tensor([[focal[0], -focal[1]]], dtype=torch.float32)
    c_t = torch.tensor([[c[0], c[1]]], dtype=torch.float32)
    return poses, focal_t, c_t, torch.tensor([float(W), float(H)])


# ---------------------------------------------------------------------------- scenes

# name -> geometry.  *_mini variants shrink the grids so golden generation on the CPU
# reference stays in seconds; the full-size ones are the BASELINE.json configs.
SCENES = {
    # BASELINE configs (1)(2): NMR 64x64, 1 source view
    "sn64": dict(W=64, H=64, NS=1, SB=1, Hl=32, Wl=32, focal=(119.4256, 119.4256), c=(32.0, 32.0),
                 z_near=1.2, z_far=4.0, radius=2.732, src=[(30.0, -20.0)], tgt=(75.0, -20.0),
                 white_bkgd=True, blender=False),
    # BASELINE config (3): SRN cars 128x128, 2 source views (S3)
    "srn_car": dict(W=128, H=128, NS=2, SB=1, Hl=64, Wl=64, focal=(131.25, 131.25), c=(64.0, 64.0),
                    z_near=0.8, z_far=1.8, radius=1.3, src=[(0.0, -20.0), (60.0, -20.0)],
                    tgt=(30.0, -20.0), white_bkgd=True, blender=True),
    "srn_mini": dict(W=32, H=32, NS=2, SB=1, Hl=16, Wl=16, focal=(32.8125, 32.8125), c=(16.0, 16.0),
                     z_near=0.8, z_far=1.8, radius=1.3, src=[(0.0, -20.0), (60.0, -20.0)],
                     tgt=(30.0, -20.0), white_bkgd=True, blender=True),
    # BASELINE config (4): DTU 400x300, 3 source views, black background (S4)
    "dtu": dict(W=400, H=300, NS=3, SB=1, Hl=150, Wl=200, focal=(723.0, 723.0), c=(200.0, 150.0),
                z_near=0.1, z_far=5.0, radius=2.0, src=[(-20.0, -15.0), (0.0, -25.0), (20.0, -15.0)],
                tgt=(0.0, -15.0), white_bkgd=False, blender=False),
    # the reference's DTU TRAINING batch (README.md:204,253: -B 4 objects, 3 source views each, 128 rays per object): twelve full-size
    # grids (737 MB); the four objects see the same camera rig (target_rays() turns the target camera by 40 degrees per object)
    "dtu_train4": dict(W=400, H=300, NS=3, SB=4, Hl=150, Wl=200, focal=(723.0, 723.0), c=(200.0, 150.0),
                       z_near=0.1, z_far=5.0, radius=2.0, src=[(-20.0, -15.0), (0.0, -25.0), (20.0, -15.0)] * 4,
                       tgt=(0.0, -15.0), white_bkgd=False, blender=False),
    "dtu_mini": dict(W=40, H=30, NS=3, SB=1, Hl=15, Wl=20, focal=(72.3, 70.1), c=(20.5, 14.25),
                     z_near=0.1, z_far=5.0, radius=2.0,
                     src=[(-20.0, -15.0), (0.0, -25.0), (20.0, -15.0)],
                     tgt=(0.0, -15.0), white_bkgd=False, blender=False),
    # the reference's 6- and 9-view DTU evaluations (README.md:201-202: `-P '22 25 28 40 44 48'` / 9 views; models.py:102-105
    # num_views_per_obj, resnetfc.py:168-172 the pooled mean over 6 / 9 rows): dtu_mini's geometry (black background, fx != fy,
    # off-centre principal point) seen from a ring of source cameras, some of them steep enough that part of the object projects
    # outside their image (border-clamped lookups in several views at once)
    "dtu6_mini": dict(W=40, H=30, NS=6, SB=1, Hl=15, Wl=20, focal=(72.3, 70.1), c=(20.5, 14.25),
                      z_near=0.1, z_far=5.0, radius=2.0,
                      src=[(-50.0, -15.0), (-30.0, -35.0), (-10.0, -10.0), (10.0, -30.0), (30.0, -5.0), (50.0, -20.0)],
                      tgt=(0.0, -15.0), white_bkgd=False, blender=False),
    "dtu9_mini": dict(W=40, H=30, NS=9, SB=1, Hl=15, Wl=20, focal=(72.3, 70.1), c=(20.5, 14.25),
                      z_near=0.1, z_far=5.0, radius=2.0,
                      src=[(-80.0, -15.0), (-60.0, -35.0), (-40.0, -10.0), (-20.0, -30.0), (0.0, -5.0), (20.0, -25.0),
                           (40.0, -12.0), (60.0, -40.0), (80.0, -20.0)],
                      tgt=(0.0, -15.0), white_bkgd=False, blender=False),
    # full DTU grid under the 9-view recipe: 9 x 512 x 150 x 200 fp32 = 553 MB, folded tables 1.66 GB per network
    "dtu_9v": dict(W=400, H=300, NS=9, SB=1, Hl=150, Wl=200, focal=(723.0, 723.0), c=(200.0, 150.0),
                   z_near=0.1, z_far=5.0, radius=2.0,
                   src=[(-80.0, -15.0), (-60.0, -35.0), (-40.0, -10.0), (-20.0, -30.0), (0.0, -5.0), (20.0, -25.0),
                        (40.0, -12.0), (60.0, -40.0), (80.0, -20.0)],
                   tgt=(0.0, -15.0), white_bkgd=False, blender=False),
    # DTU-style training (README.md:204 "in training, we always provide 3-views"; train/train.py:138-160 --nviews): 2 objects x 3
    # source views, black background, fx != fy -- the backward of the 3-row view mean with object-major rows
    "train_mv3": dict(W=40, H=30, NS=3, SB=2, Hl=15, Wl=20, focal=(72.3, 70.1), c=(20.5, 14.25),
                      z_near=0.1, z_far=5.0, radius=2.0,
                      src=[(-20.0, -15.0), (0.0, -25.0), (20.0, -15.0), (100.0, -20.0), (125.0, -10.0), (150.0, -30.0)],
                      tgt=(0.0, -15.0), white_bkgd=False, blender=False),
    # BASELINE config (5) geometry: 4 objects x 1 view (S5)
    "train": dict(W=64, H=64, NS=1, SB=4, Hl=32, Wl=32, focal=(119.4256, 119.4256), c=(32.0, 32.0),
                  z_near=1.2, z_far=4.0, radius=2.732,
                  src=[(30.0, -20.0), (100.0, -30.0), (200.0, -10.0), (300.0, -25.0)],
                  tgt=(75.0, -20.0), white_bkgd=True, blender=False),
    # config (5)'s step on a multi-view scene: 2 objects x 2 source views (view pooling in forward AND backward)
    "train_mv": dict(W=64, H=64, NS=2, SB=2, Hl=32, Wl=32, focal=(119.4256, 119.4256), c=(32.0, 32.0),
                     z_near=1.2, z_far=4.0, radius=2.732,
                     src=[(30.0, -20.0), (100.0, -30.0), (200.0, -10.0), (300.0, -25.0)],
                     tgt=(75.0, -20.0), white_bkgd=True, blender=False),
    # one axis-aligned source camera (pose entries exactly 0 / +-1, so points with camera-space z == 0 can be written
    # down exactly in fp32) + one ordinary view: the "on / behind the camera plane" fixtures
    "plane_mini": dict(W=32, H=32, NS=2, SB=1, Hl=16, Wl=16, focal=(59.7, 59.7), c=(16.0, 16.0),
                       z_near=0.5, z_far=3.5, radius=2.0, src=[(0.0, 0.0), (70.0, -30.0)],
                       tgt=(75.0, -20.0), white_bkgd=True, blender=False),
    # 2 objects x 2 views: exercises object-major view indexing (row = obj*NS + view)
    "mv_mini": dict(W=32, H=32, NS=2, SB=2, Hl=16, Wl=16, focal=(59.7, 59.7), c=(16.0, 16.0),
                    z_near=1.2, z_far=4.0, radius=2.732,
                    src=[(30.0, -20.0), (80.0, -30.0), (200.0, -10.0), (250.0, -25.0)],
                    tgt=(75.0, -20.0), white_bkgd=True, blender=False),
}


def make_scene(name, seed=2, latent_scale=0.5, with_latent=True):
    """Returns (scene, meta).  scene holds exactly what PixelNeRFNet.encode() leaves behind
    (SURVEY.md §3.4): latent NCHW (SB*NS,512,Hl,Wl), poses (SB*NS,3,4), focal (1,2), c (1,2),
    image_shape (2), NS, SB.  Source views are object-major: row = obj*NS + view."""
    g = SCENES[name]
    rs = np.random.RandomState(seed)
    NV = g["SB"] * g["NS"]
    # (with_latent=False: timing-only callers of the largest scenes draw the grid on the device themselves -- 737 MB of
    # single-threaded numpy draws take 20 s)
    latent = torch.from_numpy(
        (rs.randn(NV, 512, g["Hl"], g["Wl"]) * latent_scale).astype(np.float32)) if with_latent else None
    pre = coord_from_blender() if g["blender"] else torch.eye(4)
    assert len(g["src"]) == NV
    src = torch.stack([pre @ pose_spherical(t, p, g["radius"]) for (t, p) in g["src"]], 0)
    poses, focal, c, image_shape = encode_state(src, g["focal"], g["c"], g["W"], g["H"])
    scene = dict(latent=latent, poses=poses, focal=focal, c=c, image_shape=image_shape,
                 NS=g["NS"], SB=g["SB"])
    return scene, dict(g, src_c2w=src, pre=pre)


def target_rays(meta, n_rays=None, seed=7):
    """All rays of the target view, (SB, H*W, 8) (each object sees the same target camera,
    rotated by 40 deg per object so objects differ), optionally a seeded random subset of
    n_rays per object (the reference's train-time pixel sampling, train/train.py:143-179)."""
    g = meta
    rays_all = []
    rs = np.random.RandomState(seed)
    for o in range(g["SB"]):
        t, p = g["tgt"]
        pose = g["pre"] @ pose_spherical(t + 40.0 * o, p, g["radius"])
        r = gen_rays(pose[None], g["W"], g["H"], g["focal"], g["z_near"], g["z_far"],
                     c=g["c"]).reshape(-1, 8)
        if n_rays is not None:
            idx = torch.from_numpy(rs.choice(r.shape[0], n_rays, replace=False)).long()
            r = r[idx]
        rays_all.append(r)
    return torch.stack(rays_all, 0).contiguous()


def make_noise(R, n_coarse, n_fine, n_fine_depth, seed=1234):
    """Pre-drawn random numbers in the reference's draw order (src/render/nerf.py:111,135,
    141,158).  float32, u in [0,1)."""
    rs = np.random.RandomState(seed)
    n_imp = max(n_fine - n_fine_depth, 0)

    def uni(*shape):
        u = rs.random_sample(shape).astype(np.float32)
        return torch.from_numpy(np.minimum(u, np.float32(1.0 - 2 ** -24)))

    noise = dict(u1=uni(R, n_coarse))
    if n_fine > 0:
        if n_imp > 0:
            noise["u2"] = uni(R, n_imp)
            noise["u3"] = uni(R, n_imp)
        if n_fine_depth > 0:
            noise["n4"] = torch.from_numpy(rs.randn(R, n_fine_depth).astype(np.float32))
    return noise


PYRAMIDS = {
    # name: (NV, [(C, H, W) per ResNet stage])  -- encoder output-formatting fixtures
    "pool": (2, [(64, 9, 11), (64, 5, 6), (128, 3, 3), (256, 2, 2)]),       # conv1 s2 + maxpool (srn/dtu style), odd sizes
    "nopool": (1, [(64, 12, 12), (64, 12, 12), (128, 6, 6), (256, 3, 3)]),  # use_first_pool=False (sn64): stage 1 = stage 0 size
    # full-size shapes of BASELINE config (4): DTU 300x400 image, 3 source views
    "dtu": (3, [(64, 150, 200), (64, 75, 100), (128, 38, 50), (256, 19, 25)]),
}


def pyramid_stages(name, seed=4242):
    """Seeded stand-ins for the ResNet-34 stage outputs that SpatialEncoder.forward upsamples and
    concatenates (src/model/encoder.py:128-160)."""
    NV, shapes = PYRAMIDS[name]
    rs = np.random.RandomState(seed)
    return [torch.from_numpy(rs.randn(NV, c, h, w).astype(np.float32)) for c, h, w in shapes]


GRAD_SAMPLES = 2048


def _name_hash(s):
    h = 0
    for ch in s:
        h = (h * 131 + ord(ch)) % (2 ** 31 - 1)
    return h


def grad_sample_index(numel, key):
    """Seeded positions at which tests/golden/gradients.npz freezes a gradient tensor of the reference
    (the whole tensor when it has at most GRAD_SAMPLES elements)."""
    if numel <= GRAD_SAMPLES:
        return np.arange(numel)
    rs = np.random.RandomState(_name_hash(key))
    return np.sort(rs.choice(numel, GRAD_SAMPLES, replace=False))


# ---------------------------------------------------------------------------------------------------------------
# model configurations OUTSIDE the one every shipped experiment resolves to (src/model/models.py:22-65 accepts them all; the
# reference's default for use_code_viewdirs is even True): the composed forward's fixtures (tests/golden/variants.npz, frozen from
# the unmodified reference by oracle/make_goldens.py `variants`).  name -> (scene, model overrides, mlp conf)
VARIANTS = {
    # the reference's DEFAULT code arrangement: positions and view directions coded together (d_in = 6 x 13 = 78), shipped MLP shape
    "code_viewdirs": ("mv_mini", dict(use_code_viewdirs=True),
                      dict(n_blocks=5, d_hidden=512, combine_layer=3, combine_type="average")),
    # camera-space (not rotation-only) positions; narrow network, views pooled after the first block
    "no_normalize_z": ("dtu_mini", dict(normalize_z=False),
                       dict(n_blocks=3, d_hidden=128, combine_layer=1, combine_type="average")),
    # depth-only position feature (use_xyz=False: -z of the camera-space point, coded: 13 + 3 raw view direction)
    "z_only": ("mv_mini", dict(use_xyz=False), dict(n_blocks=2, d_hidden=64, combine_layer=1, combine_type="average")),
    # Softplus blocks + SPADE modulation + view maximum, a width that is no multiple of any tile size
    "softplus_spade_max": ("mv_mini", dict(), dict(n_blocks=4, d_hidden=96, combine_layer=2, combine_type="max", beta=3.0,
                                                   use_spade=True)),
    # a global image latent in front of the pixel-aligned one
    "global_encoder": ("srn_mini", dict(use_global_encoder=True, global_encoder=dict(backbone="resnet34", pretrained=False, latent_size=16)),
                       dict(n_blocks=3, d_hidden=128, combine_layer=2, combine_type="average")),
    # no image features at all (d_latent = 0: no lin_z), raw xyz without code or view directions (d_in = 3), never pooled
    "no_encoder_raw_xyz": ("sn64", dict(use_encoder=False, use_code=False, use_viewdirs=False),
                           dict(n_blocks=2, d_hidden=64, combine_type="average")),
}
VARIANT_POINTS = 40
VARIANT_SEED = 31


def variant_model_conf(name):
    """the reference's model conf (conf/default.conf:3-48 keys) of one VARIANTS entry, as a plain nested dict"""
    _, over, mlp = VARIANTS[name]
    conf = dict(use_encoder=True, use_global_encoder=False, use_xyz=True, canon_xyz=False, normalize_z=True,
                use_code=True, code=dict(num_freqs=6, freq_factor=1.5, include_input=True),
                use_viewdirs=True, use_code_viewdirs=False,
                mlp_coarse=dict(mlp, type="resnet"), mlp_fine=dict(mlp, type="resnet"),
                encoder=dict(backbone="resnet34", pretrained=False, num_layers=4))
    conf.update(over)
    return conf


def fill_state(shapes, seed):
    """Seeded parameters for ANY ResnetFC: `shapes` = [(state_dict key, shape)] in state_dict order -> {key: float32 tensor}.
    Weights kaiming fan-in scaled like the reference's init (resnetfc.py:36-39,89-117) except fc_1 (zero-initialised there)
    ~ N(0, 0.03^2); biases ~ N(0, 0.01^2) -- every term of the network contributes, as in make_mlp_params."""
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape in shapes:
        shape = tuple(int(v) for v in shape)
        if len(shape) == 2:
            std = 0.03 if key.endswith("fc_1.weight") else math.sqrt(2.0 / shape[1])
            out[key] = torch.from_numpy((rs.randn(*shape) * std).astype(np.float32))
        else:
            out[key] = torch.from_numpy((rs.randn(*shape) * 0.01).astype(np.float32))
    return out


def variant_inputs(name):
    """-> scene, meta, xyz (SB, P, 3), viewdirs (SB, P, 3), global latent (SB*NS, 16) | None: seeded query points around the object
    (a few far outside, projecting off the source images)"""
    scene_name, over, _ = VARIANTS[name]
    scene, meta = make_scene(scene_name, seed=2)
    rs = np.random.RandomState(_name_hash(name) % (2 ** 31))
    SB, P = scene["SB"], VARIANT_POINTS
    xyz = rs.uniform(-1.0, 1.0, (SB, P, 3)).astype(np.float32)
    xyz[:, :4] *= 4.0
    vd = rs.randn(SB, P, 3).astype(np.float32)
    vd /= np.linalg.norm(vd, axis=-1, keepdims=True)
    glob = None
    if over.get("use_global_encoder"):
        glob = torch.from_numpy((rs.randn(SB * scene["NS"], over["global_encoder"]["latent_size"]) * 0.5).astype(np.float32))
    return scene, meta, torch.from_numpy(xyz), torch.from_numpy(vd), glob


def resnetfc_shapes(d_in, d_latent, d_hidden=128, n_blocks=5, combine_layer=1000, use_spade=False, d_out=4, **_):
    """[(state_dict key, shape)] of a ResnetFC in the reference's state_dict order (registration order of resnetfc.py:87-124;
    oracle/make_goldens.py asserts it against the reference module when it freezes the variants fixture)."""
    out = []

    def lin(name, fan_out, fan_in):
        out.extend([(name + ".weight", (fan_out, fan_in)), (name + ".bias", (fan_out,))])
    if d_in > 0:
        lin("lin_in", d_hidden, d_in)
    lin("lin_out", d_out, d_hidden)
    for b in range(n_blocks):
        lin(f"blocks.{b}.fc_0", d_hidden, d_hidden)
        lin(f"blocks.{b}.fc_1", d_hidden, d_hidden)
    if d_latent != 0:
        for name in ("lin_z",) + (("scale_z",) if use_spade else ()):
            for b in range(min(combine_layer, n_blocks)):
                lin(f"{name}.{b}", d_hidden, d_latent)
    return out


def variant_mlp_params(name, d_in, d_latent):
    """-> (coarse, fine) state dicts of a VARIANTS entry (what the fixture's reference networks were loaded with)"""
    shapes = resnetfc_shapes(d_in, d_latent, **VARIANTS[name][2])
    return fill_state(shapes, VARIANT_SEED), fill_state(shapes, VARIANT_SEED + 1)
