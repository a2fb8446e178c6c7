"""ResnetFC / ResnetBlockFC: parameter containers with the reference's names, shapes and
initialisation (src/model/resnetfc.py:10-130), so that reference checkpoints load unchanged
(`mlp_coarse.lin_in.weight`, `mlp_coarse.blocks.N.fc_0.weight`, `mlp_coarse.lin_z.N.weight`...).

The arithmetic of ResnetFC.forward (resnetfc.py:132-184) of THE shape every shipped config resolves to (42 + 512 -> 512 x 5
blocks -> 4, combine_layer 3) runs inside the fused HIP kernels (csrc/pnr_mlp.hip, pnr_split.hip); `packed(precision)` hands them
their fragment stream and re-packs whenever a parameter changed.  Every OTHER shape the reference's constructor accepts (other
widths / block counts / combine layers, Softplus blocks, SPADE, d_in = 0, d_latent = 0) is composed on the host from one HIP
operator per nn.Linear (`autograd.linear_autograd`: pnr_linear / pnr_linear_backward, ReLU and residual folded in) -- slower
than the fused chain, differentiable, same results (`_forward_composed`)."""
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
    """fc_0 / fc_1 (+ shortcut when the widths differ) with the reference's names and initialisation (resnetfc.py:19-50).
    The reference's own size_in != size_out branch cannot be constructed (it initialises the bias of a bias-free Linear,
    resnetfc.py:48-49); here the shortcut is simply the bias-free kaiming Linear that branch asks for."""

    def __init__(self, size_in, size_out=None, size_h=None, beta=0.0):
        super().__init__()
        size_out = size_in if size_out is None else size_out
        size_h = min(size_in, size_out) if size_h is None else size_h
        self.size_in, self.size_h, self.size_out = size_in, size_h, size_out
        self.beta = float(beta)
        self.fc_0 = nn.Linear(size_in, size_h)
        self.fc_1 = nn.Linear(size_h, size_out)
        nn.init.constant_(self.fc_0.bias, 0.0)
        nn.init.kaiming_normal_(self.fc_0.weight, a=0, mode="fan_in")
        nn.init.constant_(self.fc_1.bias, 0.0)
        nn.init.zeros_(self.fc_1.weight)
        self.activation = nn.Softplus(beta=beta) if beta > 0 else nn.ReLU()
        if size_in == size_out:
            self.shortcut = None
        else:
            self.shortcut = nn.Linear(size_in, size_out, bias=False)
            nn.init.kaiming_normal_(self.shortcut.weight, a=0, mode="fan_in")

    def forward(self, x, precision="f16x3"):
        """resnetfc.py:53-62 as three HIP linears: ReLU folded into the operator that consumes it, the residual into fc_1's
        (a Softplus block applies its activation as a torch op in front)."""
        from ..autograd import linear_autograd as lin
        with torch.profiler.record_function("resblock"):  # the reference's scope name (resnetfc.py:54)
            relu = self.beta <= 0
            net = lin(x if relu else self.activation(x), self.fc_0.weight, self.fc_0.bias, relu_in=relu, precision=precision)
            x_s = x if self.shortcut is None else lin(x, self.shortcut.weight, None, precision=precision)
            return lin(net if relu else self.activation(net), self.fc_1.weight, self.fc_1.bias, relu_in=relu, residual=x_s,
                       precision=precision)


class ResnetFC(nn.Module):
    def __init__(self, d_in, d_out=4, n_blocks=5, d_latent=0, d_hidden=128, beta=0.0, combine_layer=1000,
                 combine_type="average", use_spade=False):
        super().__init__()
        if d_in > 0:
            self.lin_in = nn.Linear(d_in, d_hidden)
            nn.init.constant_(self.lin_in.bias, 0.0)
            nn.init.kaiming_normal_(self.lin_in.weight, a=0, mode="fan_in")
        self.lin_out = nn.Linear(d_hidden, d_out)
        nn.init.constant_(self.lin_out.bias, 0.0)
        nn.init.kaiming_normal_(self.lin_out.weight, a=0, mode="fan_in")
        self.n_blocks, self.d_latent, self.d_in, self.d_out, self.d_hidden = n_blocks, d_latent, d_in, d_out, d_hidden
        self.combine_layer, self.combine_type, self.use_spade = combine_layer, combine_type, use_spade
        self.beta = float(beta)
        self.blocks = nn.ModuleList([ResnetBlockFC(d_hidden, beta=beta) for _ in range(n_blocks)])
        if d_latent != 0:
            n_lin_z = min(combine_layer, n_blocks)

            def per_block():
                mods = nn.ModuleList([nn.Linear(d_latent, d_hidden) for _ in range(n_lin_z)])
                for m in mods:
                    nn.init.constant_(m.bias, 0.0)
                    nn.init.kaiming_normal_(m.weight, a=0, mode="fan_in")
                return mods
            self.lin_z = per_block()
            if use_spade:
                self.scale_z = per_block()
        self.activation = nn.Softplus(beta=beta) if beta > 0 else nn.ReLU()
        self.composed_precision = "f16x3"  # arithmetic of the per-Linear operators of a non-shipped shape ('f16x3' | 'f32')
        self._packed = {}

    def supported(self):
        """The one shape the fused kernel implements = the one shape the reference ships."""
        return (self.d_in == 42 and self.d_out == 4 and self.n_blocks == 5 and self.d_latent == 512
                and self.d_hidden == 512 and self.combine_layer == 3 and self.combine_type in ("average", "max")
                and not self.use_spade and self.beta <= 0)

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

    def packed(self, precision="f16", folded=False, training_pass=False):
        """training_pass: the caller is the differentiable path (autograd.py) -- see _cached"""
        if not self.supported():
            raise NotImplementedError(
                "fused HIP network supports d_in=42, d_latent=512, d_hidden=512, n_blocks=5, "
                "combine_layer=3, combine_type average | max (conf/default_mv.conf); got a different ResnetFC")
        return self._cached((precision, bool(folded)), precision,
                            lambda out: ops.pack_mlp(None, precision, folded=folded, weights=self._wstruct(), out=out), training_pass)

    def _cached(self, key, precision, build, training_pass=False):
        """entry = [fingerprint, stream, content fingerprint recorded?, times served from the cache].
        A training loop re-packs on every call (the optimizer-step count is part of the fingerprint) and never serves a stream
        twice: there the device-side content fingerprint (pnr_params_checksum, ~25 us per network per step) protects nothing, so
        a stream that replaces one that was never re-used is packed WITHOUT it -- but only while gradients are being taken: a
        stream packed under no_grad / in eval mode (a validation pass between optimizer steps) always gets its fingerprint.
        "Gradients are being taken" = the differentiable path says so (training_pass=True from autograd.py: its kernels run INSIDE
        torch.autograd.Function.forward / backward, where torch.is_grad_enabled() is False -- read there, the flag made every
        training step fingerprint every stream it packed: 4 x 21 us per step, round 6's sessions until r06_s24) or, for any other
        caller, train mode with grad enabled.
        The first call that RE-USES a stream packed without a fingerprint packs it again (and re-folds the tables), recording one (recording
        the checksum of the current parameters instead would bless a stream that a `.data` write has already made stale -- e.g. a
        validation forward with EMA weights swapped in and back through `p.data.copy_` right after `opt.step`).  That costs one
        extra pack on the first step of an mlp_fine=None training run only: from the second step on the replaced stream HAS been
        re-used, so its successor is packed with a fingerprint and every hit is verified as described above."""
        fp = self._fingerprint()
        ent = self._packed.get(key)
        checked = precision != "f32"
        fresh = ent is not None and ent[0] == fp
        if fresh and checked:
            if not ent[2]:
                # never verified against the parameters: cannot be trusted on a re-use.  Drop it AND what depends on the
                # fingerprint (the folded lin_z tables): the epoch bump re-folds them from the live parameters too
                self.invalidate_packed()
                fp, ent, fresh = self._fingerprint(), None, False
            elif self._content_verify(key):
                fp, ent, fresh = self._fingerprint(), None, False
        if not fresh:
            # the previous stream's buffer is overwritten in place (its users are earlier launches on the same stream)
            taking_grads = training_pass or (self.training and torch.is_grad_enabled())
            record = checked and (ent is None or ent[3] > 0 or ent[0] == fp or not taking_grads)
            ent = [fp, build(None if ent is None else ent[1]), record, 0]
            self._packed[key] = ent
            if record:
                self._content_record(key)
        else:
            ent[3] += 1
        return ent[1]

    def packed_bwd(self, precision="f16"):
        """transposed weight streams for the backward data-gradient chain (training: always the differentiable path)."""
        return self._cached(("bwd", precision), precision,
                            lambda out: ops.pack_mlp(None, precision, backward=True, weights=self._wstruct(), out=out), True)

    def forward(self, zx, combine_inner_dims=(1,), combine_index=None, dim_size=None):
        """src/model/resnetfc.py:132-184 on explicit rows zx (..., d_latent + d_in).  The renderer never comes here for the
        shipped model (PixelNeRFNet.forward runs the fused kernel, which also does the feature lookup); this entry serves callers
        that hold their own (z, x) rows, and PixelNeRFNet's composed forward for non-shipped model variants.
        Shipped shape under no_grad: the exact-fp32 HIP chain (pnr_resnetfc_forward_f32).  Anything else -- another shape, or
        autograd through the call -- is composed from one HIP operator per nn.Linear (differentiable)."""
        if combine_index is not None:
            raise NotImplementedError("combine_index (frustum culling) is commented out in the reference as well")
        assert zx.size(-1) == self.d_latent + self.d_in
        dims = tuple(int(d) for d in combine_inner_dims)
        with torch.profiler.record_function("resnetfc_infer"):  # the reference's scope name (resnetfc.py:141)
            wants_grad = torch.is_grad_enabled() and (zx.requires_grad or self.any_requires_grad())
            if self.supported() and not wants_grad and (dims == (1,) or len(dims) == 2):
                flat = zx.reshape(-1, zx.shape[-1])
                out = ops.resnetfc_forward(dict(self.state_dict()), flat, dims, combine_max=self._combine_max())
                if dims == (1,):
                    return out.reshape(*zx.shape[:-1], self.d_out)
                # util.combine_interleaved: (-1, NS, B, ...) pooled over dim 1 -> (-1, B, ...)   util.py:461-471
                return out.reshape(-1, dims[1], *zx.shape[1:-1], self.d_out)
            return self._forward_composed(zx, dims)

    def _forward_composed(self, zx, dims, parts=None):
        """resnetfc.py:141-184 for any constructor arguments: every nn.Linear is one `linear_autograd` node (pnr_linear /
        pnr_linear_backward; `x + lin_z(z)` rides as that operator's residual, ReLUs as its input activation); the view pooling,
        SPADE's product and a Softplus are torch ops on HIP tensors in between.
        parts = (z, x): the two column groups of zx handed over separately (PixelNeRFNet's composed forward: saves the
        concatenation and the two slice copies of a (rows, d_latent + d_in) tensor)."""
        from .. import util
        from ..autograd import linear_autograd as lin
        prec = self.composed_precision
        relu = self.beta <= 0
        if parts is not None:
            z, x = parts
            zx = x if x is not None else z
        if not zx.is_cuda:
            raise ops._lib.PixelNerfHipError("ResnetFC.forward: tensors must live on a HIP device (no CPU path)")
        if parts is None:
            zx = zx.float()
            z = zx[..., : self.d_latent] if self.d_latent > 0 else None
            x = zx[..., self.d_latent:] if self.d_latent > 0 else zx
        if self.d_in > 0:
            x = lin(x, self.lin_in.weight, self.lin_in.bias, precision=prec)
        else:
            x = torch.zeros(self.d_hidden, device=zx.device)          # resnetfc.py:149 (broadcast against the first lin_z)
        for blkid in range(self.n_blocks):
            if blkid == self.combine_layer:
                x = util.combine_interleaved(x, dims, self.combine_type)
            if self.d_latent > 0 and blkid < self.combine_layer:
                lz = self.lin_z[blkid]
                if self.use_spade:
                    x = lin(z, self.scale_z[blkid].weight, self.scale_z[blkid].bias, precision=prec) * x \
                        + lin(z, lz.weight, lz.bias, precision=prec)
                elif x.shape[:-1] == z.shape[:-1]:
                    x = lin(z, lz.weight, lz.bias, residual=x, precision=prec)
                else:
                    x = x + lin(z, lz.weight, lz.bias, precision=prec)
            x = self.blocks[blkid](x, precision=prec)
        return lin(x if relu else self.activation(x), self.lin_out.weight, self.lin_out.bias, relu_in=relu, precision=prec)

    @classmethod
    def from_conf(cls, conf, d_in, **kwargs):
        return cls(d_in, n_blocks=conf.get_int("n_blocks", 5), d_hidden=conf.get_int("d_hidden", 128),
                   beta=conf.get_float("beta", 0.0), combine_layer=conf.get_int("combine_layer", 1000),
                   combine_type=conf.get_string("combine_type", "average"),
                   use_spade=conf.get_bool("use_spade", False), **kwargs)
