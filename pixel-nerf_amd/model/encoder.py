"""SpatialEncoder: the ResNet-34 image encoder stays in PyTorch-ROCm (BASELINE north_star).
torchvision is not available here, so the trunk is written out in plain torch with
torchvision's parameter names (`model.conv1`, `model.bn1`, `model.layerN.i.convK` ...), which
keeps reference checkpoints loadable (src/model/encoder.py:13-178).

forward() builds the (SB*NS, 512, Hl, Wl) feature pyramid exactly as the reference does
(encoder.py:111-164); the per-sample bilinear lookup `index()` (encoder.py:80-109) is NOT
called by the product path -- it is fused into the HIP network kernel, which reads the
channel-last copy of `latent` made by `latent_nhwc()`."""
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
    use_graph = True
    MAX_GRAPHS = 4

    def _graphable(self, x):
        return (self.use_graph and not self.training and not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32
                and x.device == self.latent.device and self.upsample_interp == "bilinear" and not torch.cuda.is_current_stream_capturing())

    def _forward_graph(self, x):
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        key = (tuple(x.shape), x.device.index)
        ent = self._graphs.get(key)
        if ent is None:
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
        """encoder.py:111-164."""
        if self._graphable(x):
            out = self._forward_graph(x)
            if out is not None:
                return out
        return self._forward_eager(x)

    def _forward_eager(self, x, scaling=True):
        if self.feature_scale != 1.0:
            x = F.interpolate(x, scale_factor=self.feature_scale,
                              mode="bilinear" if self.feature_scale > 1.0 else "area",
                              align_corners=True if self.feature_scale > 1.0 else None,
                              recompute_scale_factor=True)
        x = x.to(device=self.latent.device)
        x = self.model.relu(self.model.bn1(self.model.conv1(x)))
        latents = [x]
        if self.num_layers > 1:
            if self.use_first_pool:
                x = self.model.maxpool(x)
            x = self.model.layer1(x)
            latents.append(x)
        if self.num_layers > 2:
            x = self.model.layer2(x)
            latents.append(x)
        if self.num_layers > 3:
            x = self.model.layer3(x)
            latents.append(x)
        if self.num_layers > 4:
            x = self.model.layer4(x)
            latents.append(x)
        self.latents = latents
        if (not torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and self.upsample_interp == "bilinear"
                and all(t.shape[1] % 64 == 0 for t in latents)):
            # inference: one HIP pass writes the NHWC grid the fused kernel reads AND the reference's NCHW tensor
            nhwc, self.latent = ops.pyramid_to_latent(latents, want_nchw=True)
            self._nhwc = ((self.latent.data_ptr(), self.latent._version, tuple(self.latent.shape)), nhwc)
            if scaling:
                self._set_scaling()
            return self.latent
        align_corners = None if self.index_interp == "nearest " else True
        latent_sz = latents[0].shape[-2:]
        for i in range(len(latents)):
            latents[i] = F.interpolate(latents[i], latent_sz, mode=self.upsample_interp, align_corners=align_corners)
        self.latent = torch.cat(latents, dim=1)
        self._set_scaling()
        return self.latent

    def _set_scaling(self):
        """encoder.py:161-163."""
        self.latent_scaling[0] = self.latent.shape[-1]
        self.latent_scaling[1] = self.latent.shape[-2]
        self.latent_scaling = self.latent_scaling / (self.latent_scaling - 1) * 2.0

    def latent_nhwc(self):
        """Channel-last copy of `latent` for the fused kernel (one bilinear corner = one
        contiguous 2 KiB row); cached until `latent` is replaced or modified."""
        lat = self.latent
        key = (lat.data_ptr(), lat._version, tuple(lat.shape))
        if self._nhwc is None or self._nhwc[0] != key:
            self._nhwc = (key, ops.nchw_to_nhwc(lat.detach().float()))
        return self._nhwc[1]

    def index(self, uv, cam_z=None, image_size=(), z_bounds=None):
        """encoder.py:80-109, for callers that use the encoder on its own (the renderer does
        not: the lookup is fused into the network kernel)."""
        with torch.profiler.record_function("encoder_index"):  # encoder.py:90
            return self._index(uv, image_size)

    def _index(self, uv, image_size):
        if uv.shape[0] == 1 and self.latent.shape[0] > 1:
            uv = uv.expand(self.latent.shape[0], -1, -1)
        if len(image_size) > 0:
            if len(image_size) == 1:
                image_size = (image_size, image_size)
            scale = self.latent_scaling / image_size
            uv = uv * scale - 1.0
        uv = uv.unsqueeze(2)
        samples = F.grid_sample(self.latent, uv, align_corners=True, mode=self.index_interp,
                                padding_mode=self.index_padding)
        return samples[:, :, :, 0]

    @classmethod
    def from_conf(cls, conf):
        return cls(conf.get_string("backbone"), pretrained=conf.get_bool("pretrained", True),
                   num_layers=conf.get_int("num_layers", 4), index_interp=conf.get_string("index_interp", "bilinear"),
                   index_padding=conf.get_string("index_padding", "border"),
                   upsample_interp=conf.get_string("upsample_interp", "bilinear"),
                   feature_scale=conf.get_float("feature_scale", 1.0),
                   use_first_pool=conf.get_bool("use_first_pool", True))
