"""SpatialEncoder: the ResNet-34 image encoder stays in PyTorch-ROCm (BASELINE north_star).
torchvision is not available here, so the trunk is written out in plain torch with
torchvision's parameter names (`model.conv1`, `model.bn1`, `model.layerN.i.convK` ...), which
keeps reference checkpoints loadable (src/model/encoder.py:13-178).

forward() yields the (SB*NS, 512, Hl, Wl) feature grid of the reference (encoder.py:111-164): the trunk's stage outputs,
each resampled to the first stage's resolution, concatenated along the channels.  The per-sample bilinear lookup `index()`
(encoder.py:80-109) is NOT called by the product path -- it is fused into the HIP network kernel, which reads the
channel-last copy of `latent` made by `latent_nhwc()`; called on its own it is a HIP operator too (`pnr_grid_index` +
backward, csrc/pnr_encode.hip), like `PositionalEncoding`."""
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from .. import ops


class _BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = norm_layer(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), norm_layer(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class _ResNet(nn.Module):
    """torchvision.models.resnet{18,34} layout (BasicBlock)."""

    def __init__(self, layers, norm_layer=nn.BatchNorm2d):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inpl = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            blocks = []
            for j in range(n):
                blocks.append(_BasicBlock(inpl, planes, (1 if i == 0 else 2) if j == 0 else 1, norm_layer))
                inpl = planes
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.avgpool = nn.Sequential()  # encoder.py:66-67
        self.fc = nn.Sequential()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _GridIndex(torch.autograd.Function):
    """`latent` (NV,C,Hl,Wl) sampled at normalised `uv` (NV,N,2): pnr_grid_index on the channel-last copy of the grid, with
    pnr_grid_index_backward behind it (gradients to the grid and to the coordinates, like F.grid_sample's)."""

    @staticmethod
    def forward(ctx, latent, uv, enc):
        nhwc = enc.latent_nhwc() if latent is enc.latent else ops.nchw_to_nhwc(latent.detach().float())
        uv = uv.contiguous()
        ctx.save_for_backward(nhwc, uv)
        return ops.grid_index(nhwc, uv)

    @staticmethod
    def backward(ctx, g):
        nhwc, uv = ctx.saved_tensors
        d_lat, d_uv = ops.grid_index_backward(nhwc, uv, g.contiguous().float(), want_latent=ctx.needs_input_grad[0],
                                              want_uv=ctx.needs_input_grad[1])
        return (None if d_lat is None else d_lat.permute(0, 3, 1, 2)), d_uv, None


class SpatialEncoder(nn.Module):
    def __init__(self, backbone="resnet34", pretrained=True, num_layers=4, index_interp="bilinear",
                 index_padding="border", upsample_interp="bilinear", feature_scale=1.0, use_first_pool=True,
                 norm_type="batch"):
        super().__init__()
        if backbone not in ("resnet34", "resnet18"):
            raise NotImplementedError("only the resnet18/34 backbones of the shipped configs (encoder.py:56)")
        if norm_type != "batch":
            raise NotImplementedError("norm_type != batch is not used by any shipped config")
        # `pretrained` ImageNet weights cannot be fetched offline (torchvision and the network are absent): take them
        # from a local torchvision-format state_dict (PIXELNERF_RESNET_WEIGHTS=/path/resnet34.pth) or load a
        # pixelNeRF checkpoint (state_dict keys encoder.model.*) afterwards; say so instead of silently starting from
        # a random trunk where the reference would start from ImageNet features.
        self.feature_scale = feature_scale
        self.use_first_pool = use_first_pool
        self.model = _ResNet([3, 4, 6, 3] if backbone == "resnet34" else [2, 2, 2, 2])
        self.latent_size = [0, 64, 128, 256, 512, 1024][num_layers]
        self.num_layers = num_layers
        self.index_interp, self.index_padding, self.upsample_interp = index_interp, index_padding, upsample_interp
        self.register_buffer("latent", torch.empty(1, 1, 1, 1), persistent=False)
        self.register_buffer("latent_scaling", torch.empty(2, dtype=torch.float32), persistent=False)
        self._nhwc = None
        if pretrained:
            import os
            path = os.environ.get("PIXELNERF_RESNET_WEIGHTS")
            if path and os.path.exists(path):
                sd = torch.load(path, map_location="cpu")
                missing = self.model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("fc.")}, strict=False)
                if missing.missing_keys:
                    warnings.warn(f"SpatialEncoder: {path} lacks {len(missing.missing_keys)} trunk tensors")
            else:
                warnings.warn("SpatialEncoder(pretrained=True): no ImageNet weights available offline -- the ResNet trunk is "
                              "randomly initialised until a checkpoint is loaded (set PIXELNERF_RESNET_WEIGHTS to a "
                              "torchvision resnet state_dict to reproduce the reference's initialisation)")

    # Inference encodes are launch-bound (~100 small kernels for a 64x64 image: 1.6 ms of which < 0.3 ms is GPU work): in eval
    # mode under no_grad the trunk + the formatting pass are captured ONCE per input shape into a HIP graph and replayed
    # (torch.cuda.CUDAGraph; the pixelnerf_amd kernels launch on the capturing stream like any other).  The graph's output
    # buffers are cloned, so every call still returns fresh tensors like the reference.  Any failure while capturing turns
    # the feature off for the module (plain eager launches).  SpatialEncoder.use_graph = False disables it.
    # A captured graph holds raw pointers to the trunk's parameters and buffers: the cache is keyed by their addresses as well
    # (a replaced storage -> a new capture), dropped by every `_apply` (.to() / .half() / .cpu()), and never copied or pickled
    # (`__getstate__`, which copy.deepcopy and pickle both go through: a CUDAGraph is a process-local handle).
    use_graph = True
    MAX_GRAPHS = 4
    _TRANSIENT = ("_graphs", "latents", "_nhwc", "_scaling_cache")  # per-process caches: rebuilt on demand, never part of a copy / checkpoint

    def __getstate__(self):
        state = self.__dict__.copy()
        for k in self._TRANSIENT:
            state.pop(k, None)
        state["_nhwc"] = None
        return state

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop("_scaling_cache", None)
        self.__dict__.pop("_graphs", None)  # the captured launches point at the storages `fn` is about to replace
        return super()._apply(fn, *args, **kwargs)

    def _storage_fingerprint(self):
        return tuple(t.data_ptr() for t in list(self.model.parameters()) + list(self.model.buffers()))

    def _graphable(self, x):
        return (self.use_graph and not self.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
                and x.device == self.latent.device and self.upsample_interp == "bilinear" and not torch.cuda.is_current_stream_capturing())

    def _forward_graph(self, x):
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        key = (tuple(x.shape), x.device.index, self._storage_fingerprint())
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= self.MAX_GRAPHS:
                stale = [k for k in self._graphs if k[2] != key[2]]  # captures of storages that no longer exist
                for k in stale:
                    del self._graphs[k]
                if len(self._graphs) >= self.MAX_GRAPHS:
                    return None
            try:
                static_in = x.clone()
                side = torch.cuda.Stream(device=x.device)
                side.wait_stream(torch.cuda.current_stream(x.device))
                with torch.cuda.stream(side):  # warm-up off the capture (MIOpen picks its algorithms, allocator grows)
                    for _ in range(2):
                        self._forward_eager(static_in, scaling=False)
                torch.cuda.current_stream(x.device).wait_stream(side)
                g = torch.cuda.CUDAGraph()
                # thread-local capture mode: other host threads (a DataLoader's pin-memory thread) may keep calling the runtime
                with torch.cuda.graph(g, capture_error_mode="thread_local"):  # (latent_scaling: host scalars, set outside the capture)
                    out = self._forward_eager(static_in, scaling=False)
                ent = (g, static_in, out, self._nhwc[1], list(self.latents))
                self._graphs[key] = ent
            except Exception as e:  # noqa: BLE001 -- whatever the capture trips over, eager launches remain correct
                warnings.warn(f"SpatialEncoder: HIP-graph capture of the encoder failed ({type(e).__name__}: {e}); using eager launches")
                type(self).use_graph = False
                self._graphs = {}
                return None
        g, static_in, out, nhwc, latents = ent
        static_in.copy_(x)
        g.replay()
        self.latent = out.clone()
        self._nhwc = ((self.latent.data_ptr(), self.latent._version, tuple(self.latent.shape)), nhwc.clone())
        self.latents = latents  # the trunk's stage outputs live in the graph's buffers (valid until the next encode of this shape)
        self._set_scaling()
        return self.latent

    def forward(self, x):
        """images (NV,3,H,W) -> `latent` (NV, latent_size, Hl, Wl), also kept on the module (encoder.py:111-164)."""
        if self._graphable(x):
            out = self._forward_graph(x)
            if out is not None:
                return out
        return self._forward_eager(x)

    def _resize_input(self, x):
        """optional input rescaling in front of the trunk (encoder.py:118-126): enlarging interpolates, shrinking averages"""
        if self.feature_scale == 1.0:
            return x
        grow = self.feature_scale > 1.0
        return F.interpolate(x, scale_factor=self.feature_scale, mode="bilinear" if grow else "area",
                             align_corners=True if grow else None, recompute_scale_factor=True)

    def _stages(self, x):
        """the stem and the first num_layers - 1 residual stages of the trunk; every stage's output is one pyramid level"""
        trunk = self.model
        levels = [trunk.relu(trunk.bn1(trunk.conv1(x)))]
        for depth, stage in enumerate((trunk.layer1, trunk.layer2, trunk.layer3, trunk.layer4)[: self.num_layers - 1]):
            feed = trunk.maxpool(levels[-1]) if (depth == 0 and self.use_first_pool) else levels[-1]
            levels.append(stage(feed))
        return levels

    def _forward_eager(self, x, scaling=True):
        levels = self._stages(self._resize_input(x).to(device=self.latent.device))
        self.latents = levels
        fused_ok = (not torch.is_grad_enabled() and x.is_cuda and levels[0].dtype == torch.float32
                    and self.upsample_interp == "bilinear" and all(t.shape[1] % 64 == 0 for t in levels))
        if fused_ok:
            # inference: one HIP pass writes the NHWC grid the fused kernel reads AND the reference's NCHW tensor
            nhwc, self.latent = ops.pyramid_to_latent(levels, want_nchw=True)
            self._nhwc = ((self.latent.data_ptr(), self.latent._version, tuple(self.latent.shape)), nhwc)
        else:
            # training (autograd must see the resampling) or a non-default interpolation: torch's own resampling of every
            # level to the first level's size.  The reference asks for align_corners=True whatever the mode (its test for
            # "nearest" can never match, encoder.py:152), which only the interpolating modes accept -- same here.
            size = levels[0].shape[-2:]
            interpolating = self.upsample_interp in ("linear", "bilinear", "bicubic", "trilinear")
            self.latent = torch.cat([F.interpolate(t, size, mode=self.upsample_interp, align_corners=True if interpolating else None)
                                     for t in levels], dim=1)
        if scaling or not fused_ok:
            self._set_scaling()
        return self.latent

    def _set_scaling(self):
        """pixel -> normalised-coordinate factors of the grid, (W, H) / (W - 1, H - 1) * 2 (encoder.py:161-163)"""
        # a pure function of the grid size: built once per (size, device) -- a host -> device copy per encode would put a
        # synchronising transfer into an otherwise launch-only (HIP-graph replayed) inference encode
        key = (int(self.latent.shape[-1]), int(self.latent.shape[-2]), str(self.latent_scaling.device))
        cache = self.__dict__.setdefault("_scaling_cache", {})
        if key not in cache:
            wh = torch.tensor([float(key[0]), float(key[1])], dtype=torch.float32, device=self.latent_scaling.device)
            cache[key] = wh / (wh - 1.0) * 2.0
        self.latent_scaling = cache[key].clone()

    def latent_nhwc(self):
        """Channel-last copy of `latent` for the fused kernel (one bilinear corner = one
        contiguous 2 KiB row); cached until `latent` is replaced or modified."""
        lat = self.latent
        key = (lat.data_ptr(), lat._version, tuple(lat.shape))
        if self._nhwc is None or self._nhwc[0] != key:
            self._nhwc = (key, ops.nchw_to_nhwc(lat.detach().float()))
        return self._nhwc[1]

    def index(self, uv, cam_z=None, image_size=(), z_bounds=None):
        """Pixel-aligned features at image points (encoder.py:80-109), for callers that use the encoder on its own (the
        renderer does not: the lookup is fused into the network kernel).
        :param uv (NV|1, N, 2) image points (x, y); pixel units when `image_size` is given, else already in [-1, 1]
        :param image_size () | (s,) | (w, h): the image extent the pixel coordinates refer to
        :return (NV, latent_size, N)"""
        with torch.profiler.record_function("encoder_index"):  # the reference's scope name (encoder.py:90)
            lat = self.latent
            if uv.shape[0] == 1 and lat.shape[0] > 1:
                uv = uv.expand(lat.shape[0], -1, -1)
            if len(image_size) > 0:
                if torch.is_tensor(image_size):
                    extent = image_size.to(device=uv.device, dtype=torch.float32).reshape(-1)
                else:
                    extent = torch.tensor([float(v) for v in image_size], dtype=torch.float32, device=uv.device)
                if extent.numel() == 1:
                    extent = extent.expand(2)
                uv = uv * (self.latent_scaling / extent) - 1.0
            if self.index_interp != "bilinear" or self.index_padding != "border":
                # the HIP lookup implements what every shipped config uses (bilinear / border); any other mode the reference's
                # constructor accepts (encoder.py:27-28) is ATen's grid_sample on the HIP tensors, exactly the reference's call
                # (encoder.py:100-108) -- PixelNeRFNet.fused_supported() sends such a conf down the composed forward
                samples = F.grid_sample(lat, uv.unsqueeze(2), align_corners=True, mode=self.index_interp,
                                        padding_mode=self.index_padding)
                return samples[:, :, :, 0]
            return _GridIndex.apply(lat, uv.float(), self)

    @classmethod
    def from_conf(cls, conf):
        return cls(conf.get_string("backbone"), pretrained=conf.get_bool("pretrained", True),
                   num_layers=conf.get_int("num_layers", 4), index_interp=conf.get_string("index_interp", "bilinear"),
                   index_padding=conf.get_string("index_padding", "border"),
                   upsample_interp=conf.get_string("upsample_interp", "bilinear"),
                   feature_scale=conf.get_float("feature_scale", 1.0),
                   use_first_pool=conf.get_bool("use_first_pool", True))


class ImageEncoder(nn.Module):
    """Global image encoder (src/model/encoder.py:166-233; `use_global_encoder` of PixelNeRFNet, unused by the shipped configs):
    the same plain-torch ResNet trunk through its global average pool, one latent vector per image, optionally projected to
    `latent_size`.  State-dict keys as the reference's (`model.*`, `fc.*`).  Stays in PyTorch-ROCm like the spatial trunk."""

    def __init__(self, backbone="resnet34", pretrained=True, latent_size=128):
        super().__init__()
        if backbone not in ("resnet34", "resnet18"):
            raise NotImplementedError("ImageEncoder: resnet18 / resnet34 trunks (BasicBlock) only")
        if pretrained:
            warnings.warn("ImageEncoder: no ImageNet weights are available offline; the trunk is randomly initialised "
                          "(load a checkpoint)", stacklevel=2)
        self.model = _ResNet((3, 4, 6, 3) if backbone == "resnet34" else (2, 2, 2, 2))
        self.model.avgpool = nn.AdaptiveAvgPool2d((1, 1))  # torchvision's, kept by the reference (encoder.py:181-182 drop only fc)
        self.register_buffer("latent", torch.empty(1, 1), persistent=False)
        self.latent_size = latent_size
        if latent_size != 512:
            self.fc = nn.Linear(512, latent_size)

    def index(self, uv, cam_z=None, image_size=(), z_bounds=()):
        """(B, L) -> (B, L, N): the same vector at every query point (encoder.py:189-195)."""
        return self.latent.unsqueeze(-1).expand(-1, -1, uv.shape[1])

    def forward(self, x):
        """x (B, 3, H, W) -> latent (B, latent_size), also kept in `self.latent` (encoder.py:197-222)."""
        m = self.model
        x = x.to(device=self.latent.device)
        x = m.maxpool(m.relu(m.bn1(m.conv1(x))))
        x = m.layer4(m.layer3(m.layer2(m.layer1(x))))
        x = torch.flatten(m.avgpool(x), 1)
        if self.latent_size != 512:
            x = self.fc(x)
        self.latent = x
        return self.latent

    @classmethod
    def from_conf(cls, conf):
        return cls(conf.get_string("backbone"), pretrained=conf.get_bool("pretrained", True),
                   latent_size=conf.get_int("latent_size", 128))
