"""ResnetFC / ResnetBlockFC: parameter containers with the reference's names, shapes and
initialisation (src/model/resnetfc.py:10-130), so that reference checkpoints load unchanged
(`mlp_coarse.lin_in.weight`, `mlp_coarse.blocks.N.fc_0.weight`, `mlp_coarse.lin_z.N.weight`...).

The arithmetic of ResnetFC.forward (resnetfc.py:132-184) runs inside the fused HIP kernel
(csrc/pnr_mlp.hip); `packed(precision)` hands the kernel its fragment stream and re-packs
whenever a parameter changed."""
import weakref

import torch
from torch import nn
from torch.optim.optimizer import register_optimizer_step_post_hook

from .. import ops

# Fused optimizers (torch.optim.Adam(..., fused=True) and friends) update parameters in place WITHOUT bumping
# tensor._version, so (data_ptr, _version) alone cannot tell that the packed fragment streams / folded tables are stale.
# A post-step hook counts the optimizer steps PER ResnetFC: an optimizer.step() bumps only the networks that own one of
# the optimizer's parameters (membership is resolved once per optimizer and re-resolved when its parameter list changes),
# so unrelated optimizers in the process -- another model, a GAN's second network -- no longer force a re-pack / re-fold here.
_PARAM_OWNER = weakref.WeakValueDictionary()   # id(Parameter) -> owning ResnetFC (filled by ResnetFC._named)
_OPT_MEMBERS = weakref.WeakKeyDictionary()     # optimizer -> (number of parameters when resolved, [weakref(ResnetFC)], registry size)


def _count_optimizer_step(optimizer, args, kwargs):
    n = sum(len(g["params"]) for g in optimizer.param_groups)
    hit = _OPT_MEMBERS.get(optimizer)
    if hit is None or hit[0] != n or hit[2] != len(_PARAM_OWNER):
        owners = {}
        for g in optimizer.param_groups:
            for p in g["params"]:
                m = _PARAM_OWNER.get(id(p))
                if m is not None and any(q is p for _, q in m._named()):
                    owners[id(m)] = weakref.ref(m)
        hit = (n, list(owners.values()), len(_PARAM_OWNER))
        _OPT_MEMBERS[optimizer] = hit
    for r in hit[1]:
        m = r()
        if m is not None:
            m.__dict__["_opt_steps"] = m.__dict__.get("_opt_steps", 0) + 1


register_optimizer_step_post_hook(_count_optimizer_step)


class ResnetBlockFC(nn.Module):
    def __init__(self, size_in, size_out=None, size_h=None, beta=0.0):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        if size_in != size_out or beta > 0:
            raise NotImplementedError("fused kernel: 512->512 ReLU blocks only (all shipped configs)")
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)
        self.activation = nn.ReLU()
        self.shortcut = None


class ResnetFC(nn.Module):
    def __init__(self, d_in, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, beta=0.0, combine_layer=1000,
                 combine_type="average", use_spade=False):
        super().__init__()
        self.lin_in = nn.Linear(d_in, d_hidden)
        nn.init.constant_(self.lin_in.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_in.weight, a=0, mode="fan_in")
        self.lin_out = nn.Linear(d_hidden, d_out)
        nn.init.constant_(self.lin_out.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_out.weight, a=0, mode="fan_in")
        self.n_blocks, self.d_latent, self.d_in, self.d_out, self.d_hidden = n_blocks, d_latent, d_in, d_out, d_hidden
        self.combine_layer, self.combine_type, self.use_spade = combine_layer, combine_type, use_spade
        self.blocks = nn.ModuleList([ResnetBlockFC(d_hidden, beta=beta) for _ in range(n_blocks)])
        if d_latent != 0:
            n_lin_z = min(combine_layer, n_blocks)
            self.lin_z = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(n_lin_z)])
            for i in range(n_lin_z):
                nn.init.constant_(self.lin_z[i].bias, 0.0)
                nn.init.kaiming_normal_(self.lin_z[i].weight, a=0, mode="fan_in")
        self.activation = nn.ReLU()
        self._packed = {}

    def supported(self):
        """The one shape the fused kernel implements = the one shape the reference ships."""
        return (self.d_in == 42 and self.d_out == 4 and self.n_blocks == 5 and self.d_latent == 512
                and self.d_hidden == 512 and self.combine_layer == 3 and self.combine_type in ("average", "max")
                and not self.use_spade)

    def _combine_max(self):
        """util.combine_interleaved's agg_type (src/util/util.py:461-471): "average" in every shipped config; "max" is carried as
        a flag word of the packed network and applied where the kernels pool the source views (inference entries)"""
        return self.combine_type == "max"

    _CACHE_KEYS = ("_named_cache", "_ordered_cache", "_wstruct_cache", "_content")

    def __getstate__(self):
        # copies / pickles start with empty caches (the weight struct holds raw device pointers of THIS module's storage)
        d = dict(self.__dict__)
        for k in self._CACHE_KEYS:
            d.pop(k, None)
        d["_packed"] = {}
        return d

    # ---- host-side caches.  A training step re-packs both networks twice (forward + transposed streams); walking the
    # module tree for state_dict() / named_parameters() every time cost more host time than the pack kernels take on the GPU.
    def _named(self):
        """[(state_dict key, Parameter)] in registration order, cached (dropped by .to() / .cuda() / ._apply).  The cache is
        validated on every use against the live module tree (30 identity checks): a Parameter OBJECT that was replaced
        (`load_state_dict(..., assign=True)`, `mlp.lin_in.weight = nn.Parameter(...)`) drops every dependent cache."""
        c = self.__dict__.get("_named_cache")
        if c is not None:
            for (_, p), (mod, pname) in zip(c[0], c[1]):
                if mod._parameters.get(pname) is not p:
                    self.invalidate_packed()
                    c = None
                    break
        if c is None:
            named, where = [], []
            for mname, mod in self.named_modules():
                for pname, p in mod._parameters.items():
                    if p is not None:
                        named.append(((mname + "." if mname else "") + pname, p))
                        where.append((mod, pname))
                        _PARAM_OWNER[id(p)] = self
            c = (named, where)
            self.__dict__["_named_cache"] = c
        return c[0]

    def ordered_params(self, names):
        """the parameters in the order of `names` (autograd.PARAM_NAMES), cached per name list"""
        self._named()  # validates the caches against the live module tree
        c = self.__dict__.get("_ordered_cache")
        if c is None or c[0] is not names:
            d = dict(self._named())
            c = (names, [d[n] for n in names])
            self.__dict__["_ordered_cache"] = c
        return c[1]

    def _apply(self, fn, *args, **kwargs):
        for k in self._CACHE_KEYS:
            self.__dict__.pop(k, None)
        self.__dict__.get("_packed", {}).clear()
        return super()._apply(fn, *args, **kwargs)

    def any_requires_grad(self):
        return any(p.requires_grad for _, p in self._named())

    def _fingerprint(self):
        named = self._named()
        return (self.__dict__.get("_opt_steps", 0), self.__dict__.get("_epoch", 0), self.combine_type) + tuple((p.data_ptr(), p._version) for _, p in named)

    def _wstruct(self):
        """(PnrMlpWeights, tensors) of the current parameter storage: the struct holds pointers only, so it survives
        in-place updates and is rebuilt only when a parameter moved."""
        key = tuple(p.data_ptr() for _, p in self._named()) + (self._combine_max(),)
        c = self.__dict__.get("_wstruct_cache")
        if c is None or c[0] != key:
            c = (key, ops._weights_struct({k: p for k, p in self._named()}, self._combine_max()))
            self.__dict__["_wstruct_cache"] = c
        return c[1]

    def invalidate_packed(self):
        """Drop the packed streams (for parameter writes that neither bump tensor._version nor go through a
        torch.optim optimizer, e.g. a custom kernel writing through data_ptr(), or `p.data.mul_()` / `p.data.copy_()`:
        writes through `.data` do NOT bump `_version` -- see packed() for what is detected automatically)."""
        self._packed.clear()
        for k in self._CACHE_KEYS:
            self.__dict__.pop(k, None)
        self.__dict__["_epoch"] = self.__dict__.get("_epoch", 0) + 1  # part of the fingerprint: dependent caches (folded tables) follow

    # ---- content check behind the cache key.  (optimizer steps, data_ptr, _version) misses writes through `p.data`
    # (`p.data.mul_()`, `p.data.copy_()`, EMA updates written that way) and raw-pointer kernels.  Every cache HIT therefore
    # launches pnr_params_checksum: a 64-bit fingerprint of all 30 tensors (14 MB read, ~5 us) is compared ON THE DEVICE with
    # the fingerprint taken when the stream was packed; a mismatch raises a device flag that travels to pinned host memory
    # with an asynchronous copy.  No host synchronisation: the host looks at the flag on the NEXT call, warns loudly, drops
    # the caches and re-packs -- so at most the one call already in flight renders with the old weights, and the event is
    # never silent.  `invalidate_packed()` remains the way to make even that call exact.  Skipped inside HIP-graph capture.
    def _content_state(self, dev):
        st = self.__dict__.get("_content")
        if st is None or st["dev"] != dev:
            st = dict(dev=dev, ws=torch.zeros(ops._lib.load().pnr_params_checksum_ws_bytes() // 8, dtype=torch.int64, device=dev), sums={},
                      flag=torch.zeros(1, dtype=torch.int32, device=dev), flag_host=torch.zeros(1, dtype=torch.int32).pin_memory())
            self.__dict__["_content"] = st
        return st

    def _content_record(self, key):
        """fingerprint of the parameters as stream `key` is being packed (same stream: ordered with the pack kernels)"""
        import ctypes
        w, keep = self._wstruct()
        dev = keep["lin_in.weight"].device
        if torch.cuda.is_current_stream_capturing():
            return
        st = self._content_state(dev)
        if key not in st["sums"]:
            st["sums"][key] = torch.zeros(1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            ops._lib.check(ops._lib.load().pnr_params_checksum(ctypes.byref(w), ops._p(st["ws"]), ops._p(st["sums"][key]), None, None,
                                                               ops._stream()), "pnr_params_checksum")

    def _content_verify(self, key):
        """on a cache hit of stream `key`: react to a mismatch found by an earlier check, then enqueue the next check.
        -> True when the caches were just dropped (the caller re-packs)"""
        import ctypes
        import warnings
        st = self.__dict__.get("_content")
        if st is None or key not in st["sums"] or torch.cuda.is_current_stream_capturing():
            return False
        if int(st["flag_host"][0]) != 0:
            st["flag_host"].zero_()
            st["flag"].zero_()
            warnings.warn("pixelnerf_amd.ResnetFC: the parameters changed behind the packed-weight cache (a write through "
                          "`.data` or a raw pointer bumps neither tensor._version nor an optimizer step); the previous render "
                          "call may have used the old weights.  Re-packing now -- call mlp.invalidate_packed() after such writes.",
                          RuntimeWarning, stacklevel=3)
            self.invalidate_packed()
            return True
        w, keep = self._wstruct()
        with torch.cuda.device(st["dev"]):
            ops._lib.check(ops._lib.load().pnr_params_checksum(ctypes.byref(w), ops._p(st["ws"]), None, ops._p(st["sums"][key]),
                                                               ops._p(st["flag"]), ops._stream()), "pnr_params_checksum")
        st["flag_host"].copy_(st["flag"], non_blocking=True)
        return False

    def packed(self, precision="f16", folded=False):
        if not self.supported():
            raise NotImplementedError(
                "fused HIP network supports d_in=42, d_latent=512, d_hidden=512, n_blocks=5, "
                "combine_layer=3, combine_type average | max (conf/default_mv.conf); got a different ResnetFC")
        return self._cached((precision, bool(folded)), precision,
                            lambda out: ops.pack_mlp(None, precision, folded=folded, weights=self._wstruct(), out=out))

    def _cached(self, key, precision, build):
        fp = self._fingerprint()
        hit = self._packed.get(key)
        if hit is not None and hit[0] == fp and precision != "f32" and self._content_verify(key):
            fp, hit = self._fingerprint(), None
        if hit is None or hit[0] != fp:
            # the previous stream's buffer is overwritten in place (its users are earlier launches on the same stream)
            self._packed[key] = (fp, build(None if hit is None else hit[1]))
            if precision != "f32":
                self._content_record(key)
        return self._packed[key][1]

    def packed_bwd(self, precision="f16"):
        """transposed weight streams for the backward data-gradient chain (training)."""
        return self._cached(("bwd", precision), precision,
                            lambda out: ops.pack_mlp(None, precision, backward=True, weights=self._wstruct(), out=out))

    def forward(self, zx, combine_inner_dims=(1,), combine_index=None, dim_size=None):
        """src/model/resnetfc.py:132-184 on explicit rows zx (..., d_latent + d_in): the exact-fp32 HIP linears
        (pnr_resnetfc_forward_f32), inference only.  The renderer never comes here -- PixelNeRFNet.forward runs the
        fused kernel, which also does the feature lookup; this entry serves callers that hold their own (z, x) rows."""
        if not self.supported():
            raise NotImplementedError("HIP ResnetFC supports the shipped shape only (conf/default_mv.conf)")
        if combine_index is not None:
            raise NotImplementedError("combine_index (frustum culling) is commented out in the reference as well")
        if torch.is_grad_enabled() and (zx.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError(
                "autograd through a direct ResnetFC.forward call is not implemented: training goes through "
                "NeRFRenderer (train/train.py:199-215); wrap direct calls in torch.no_grad()")
        assert zx.size(-1) == self.d_latent + self.d_in
        dims = tuple(int(d) for d in combine_inner_dims)
        flat = zx.reshape(-1, zx.shape[-1])
        out = ops.resnetfc_forward(dict(self.state_dict()), flat, dims, combine_max=self._combine_max())
        if dims == (1,):
            return out.reshape(*zx.shape[:-1], self.d_out)
        # util.combine_interleaved: (-1, NS, B, ...) mean over dim 1 -> (-1, B, ...)   util.py:461-471
        return out.reshape(-1, dims[1], *zx.shape[1:-1], self.d_out)

    @classmethod
    def from_conf(cls, conf, d_in, **kwargs):
        return cls(d_in, n_blocks=conf.get_int("n_blocks", 5), d_hidden=conf.get_int("d_hidden", 128),
                   beta=conf.get_float("beta", 0.0), combine_layer=conf.get_int("combine_layer", 1000),
                   combine_type=conf.get_string("combine_type", "average"),
                   use_spade=conf.get_bool("use_spade", False), **kwargs)
