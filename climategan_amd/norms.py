"""Host-side mirror of the reference's ``climategan/norms.py`` for the hot path: SPADE and SpectralNorm.

Same class names, constructor arguments, parameter names and state-dict keys as the reference, so reference
checkpoints load unchanged; the arithmetic runs in libcgan_hip.so (HIP, gfx950) through ``ops``.
AdaptiveInstanceNorm2d / LayerNorm of the reference are dead options in the default configs (SURVEY.md
section 2a) and are not provided.
"""
import torch
import torch.nn as nn

from . import ops

DEFAULT_COMPUTE_DTYPE = torch.float16


def _grad_guard(module: nn.Module):
    """The backward kernels are not part of this build: refuse to run under autograd instead of silently
    returning tensors without a graph."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            "climategan_amd: only the forward (inference) path is implemented in HIP so far; call under "
            "torch.no_grad() (the reference's infer_all does, trainer.py:217) or set requires_grad=False")


def needs_grad(module: nn.Module, *tensors) -> bool:
    """True when this call must be recorded for autograd (grad mode on and a parameter or an input wants a gradient)."""
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters())


def _conv_train(x: ops.NHWC, weight, bias, packed, sn, stride, pad, dilation, kw) -> ops.NHWC:
    """Differentiable conv (autograd.ConvFn).  Options without a backward kernel raise."""
    from .autograd import ConvFn

    pad_mode = kw.get("pad_mode", ops.PAD_ZERO)
    if pad_mode == ops.PAD_REFLECT and kw.get("in_upsample"):
        raise NotImplementedError("climategan_amd: reflect padding on a folded upsample has no backward kernel")
    res = kw.get("residual")
    cfg = dict(c_in=x.c, stride=stride, pad=pad, dilation=dilation, act=kw.get("act", ops.ACT_NONE),
               slope=kw.get("slope", 0.2), in_upsample=bool(kw.get("in_upsample", False)),
               residual_upsample=bool(kw.get("residual_upsample", False)), pad_mode=pad_mode,
               sn_owned=bool(kw.get("sn_owned", False)), pair_in=bool(kw.get("pair_in", False)),
               want_stats=bool(kw.get("want_stats", False)))
    y_t = ConvFn.apply(x.t, weight, bias, res.t if res is not None else None, packed, cfg, sn)
    if kw.get("want_stats"):
        return ops.NHWC(y_t, weight.shape[0]), cfg.get("stats_out")   # BatchNorm statistics from the conv's epilogue, or None
    return ops.NHWC(y_t, weight.shape[0])


class _PackCache:
    """Re-pack fp32 parameters into MFMA fragment order only when they change."""

    _plain = {}          # id(cache) -> weakref(cache): the caches holding a plain conv weight (get_plain)

    def __init__(self):
        self.key = None
        self.value = None
        self._plain_src = None

    @staticmethod
    def _key_of(params, dtype):
        return tuple((p.data_ptr(), p._version, p.dtype, str(p.device)) if torch.is_tensor(p) else p
                     for p in params if p is not None) + (dtype,)

    def get(self, params, dtype, build):
        key = self._key_of(params, dtype)
        if key != self.key:
            self.value = build()
            self.key = key
        return self.value

    def get_plain(self, weight, bias, dtype):
        """The packed form of a plain conv's (weight, bias) for the training path.  An optimizer step makes EVERY such
        cache stale at once and the next forward asks for them one by one: on the first miss all stale ones (same device,
        same dtype) are re-packed together in ONE batched launch (ops.pack_conv_weights_batched; the same arithmetic as
        ops.pack_conv_weight) instead of one launch per layer -- ~200 launches per optimizer update for the Masker."""
        import weakref

        key = self._key_of((weight, bias, "plain"), dtype)
        if key == self.key:
            return self.value
        self._plain_src = (weakref.ref(weight), weakref.ref(bias) if bias is not None else None, dtype)
        _PackCache._plain[id(self)] = weakref.ref(self)
        _PackCache.repack_stale(dtype, weight.device)
        return self.value

    @staticmethod
    def repack_stale(dtype, device) -> int:
        """Re-pack, in one batched launch on the current stream, every registered plain-conv cache of ``device`` / ``dtype``
        whose parameters changed since it was filled (an optimizer step makes all of them stale at once).  The trainer
        calls it before it forks the Masker and the Painter branches of an update onto two streams: no cache is then
        touched while both branches run."""
        self = _PackCache
        stale = []
        for cid, ref in list(_PackCache._plain.items()):
            c = ref()
            src = c._plain_src if c is not None else None
            w = src[0]() if src is not None else None
            if c is None or w is None:
                del _PackCache._plain[cid]
                continue
            b = src[1]() if src[1] is not None else None
            if src[2] != dtype or w.device != device:
                continue
            k = self._key_of((w, b, "plain"), dtype)
            if k != c.key:
                stale.append((c, w, b, k))
        # buffers are overwritten in place only where the previous content was a plain pack of the same dtype made on this
        # path (key tagged "plain"): an eval-mode pack (BatchNorm folded, made by get()) may still be in use by an
        # inference call on another stream, and switching train <-> eval must not alias the two
        reuse = [c.value if (c.value is not None and c.key is not None and c.key[-1] == dtype and "plain" in c.key and
                             isinstance(c.value, ops.PackedConv)) else None for c, _, _, _ in stale]
        packed = ops.pack_conv_weights_batched([(w.data, b.data if b is not None else None) for _, w, b, _ in stale], dtype,
                                               reuse)
        for (c, _, _, k), pk in zip(stale, packed):
            c.value, c.key = pk, k
        return len(stale)


class SpectralNorm(nn.Module):
    """Spectral-norm wrapper (reference climategan/norms.py:84-143).

    State: ``module.weight_u [Cout]``, ``module.weight_v [Cin*k*k]``, ``module.weight_bar [Cout,Cin,k,k]``
    (+ ``module.bias``), all nn.Parameters with u/v ``requires_grad=False`` exactly as the reference
    (norms.py:129-139).  One power iteration runs on EVERY forward, eval included (norms.py:141-143), and
    mutates u/v in place; the convolution then uses ``w_bar / sigma``.
    """

    def __init__(self, module, name="weight", power_iterations=1):
        super().__init__()
        if power_iterations != 1:
            raise NotImplementedError("SpectralNorm: only power_iterations=1 (the reference's only use)")
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        if not hasattr(module, name + "_bar"):
            w = getattr(module, name)
            rows = w.shape[0]
            cols = w.numel() // rows
            u = torch.randn(rows)
            v = torch.randn(cols)
            u = u / (u.norm() + 1e-12)
            v = v / (v.norm() + 1e-12)
            del module._parameters[name]
            module.register_parameter(name + "_u", nn.Parameter(u, requires_grad=False))
            module.register_parameter(name + "_v", nn.Parameter(v, requires_grad=False))
            module.register_parameter(name + "_bar", nn.Parameter(w.data))

    _prepacked = None  # set for ONE use by spectral_norm_step_all() (batched power iteration + pack)

    def sn_params(self):
        m = self.module
        return (getattr(m, self.name + "_bar").data, getattr(m, self.name + "_u").data,
                getattr(m, self.name + "_v").data, m.bias.data if m.bias is not None else None)

    frozen = False      # opt-in inference mode (freeze_spectral_norm): sigma taken once, no further power iterations
    _frozen_pack = None

    def packed(self, dtype) -> ops.PackedConv:
        """Power-iterate (updates u, v) and return w_bar / sigma packed for the MFMA conv kernel."""
        if self.frozen:
            m = self.module
            w_bar = getattr(m, self.name + "_bar")
            key = (w_bar.data_ptr(), w_bar._version, dtype)
            if self._frozen_pack is None or self._frozen_pack[0] != key:
                # the ONE power iteration a forward would have run at this point; its sigma is kept from now on
                sigma = ops.spectral_norm_power_iter(w_bar.data, getattr(m, self.name + "_u").data,
                                                     getattr(m, self.name + "_v").data)
                self._sigma = sigma
                self._frozen_pack = (key, ops.pack_conv_weight(w_bar.data, m.bias.data if m.bias is not None else None,
                                                               dtype, sigma))
            self._pre_used = False
            return self._frozen_pack[1]
        pre, self._prepacked = self._prepacked, None
        self._pre_used = pre is not None and pre.dtype == dtype
        if self._pre_used:
            return pre
        m = self.module
        w_bar = getattr(m, self.name + "_bar")
        u = getattr(m, self.name + "_u")
        v = getattr(m, self.name + "_v")
        sigma = ops.spectral_norm_power_iter(w_bar.data, u.data, v.data)
        self._sigma = sigma
        return ops.pack_conv_weight(w_bar.data, m.bias.data if m.bias is not None else None, dtype, sigma)

    _sigma = None       # device scalar of the latest power iteration (needed by the backward of w_bar / sigma)
    _sn_snapshot = None  # (sigma, u, v) copies of the batched step, for ONE use by forward() (spectral_norm_step_all)
    _pre_used = False
    trainable = False   # set by modules whose whole forward has backward kernels (the discriminators)

    def forward(self, x, **conv_kwargs):
        m = self.module
        if not self.trainable:
            _grad_guard(self)
        if self.frozen and needs_grad(self, x.t):
            raise NotImplementedError("climategan_amd: a SpectralNorm frozen for inference (freeze_spectral_norm) cannot be "
                                      "trained; unfreeze it first")
        if isinstance(x, ops.PairMap):
            # split-precision inference: this forward's power iteration has run (the batched step of the decoder, or it runs
            # here); w_bar / sigma is expanded to (hi | hi | lo) and packed for this call
            pre, self._prepacked = self._prepacked, None
            if pre is None and not self.frozen:
                self._sigma = ops.spectral_norm_power_iter(getattr(m, self.name + "_bar").data,
                                                           getattr(m, self.name + "_u").data,
                                                           getattr(m, self.name + "_v").data)
            elif pre is None and self._sigma is None:
                self.packed(x.t.dtype)               # frozen: the one power iteration that fixes sigma
            pw = ops.pack_conv_weight(getattr(m, self.name + "_bar").data, m.bias.data if m.bias is not None else None,
                                      x.t.dtype, self._sigma, pair=True)
            pad = conv_kwargs.pop("pad", m.padding[0])
            return ops.conv2d(x, pw, stride=m.stride[0], pad=pad, dilation=m.dilation[0], **conv_kwargs)
        pw = self.packed(x.t.dtype)
        pad = conv_kwargs.pop("pad", m.padding[0])   # Conv2dBlock pads with a separate module (padding=0 on the conv)
        if self.trainable and needs_grad(self, x.t):
            snap, self._sn_snapshot = self._sn_snapshot, None         # consumed once, like the pre-packed weight
            if self._pre_used and snap is not None:
                sn = snap
                conv_kwargs = dict(conv_kwargs, sn_owned=True)        # already private copies: ConvFn need not clone
            else:
                sn = (self._sigma, getattr(m, self.name + "_u").data, getattr(m, self.name + "_v").data)
            return _conv_train(x, getattr(m, self.name + "_bar"), m.bias, pw, sn, m.stride[0], pad, m.dilation[0],
                               conv_kwargs)
        return ops.conv2d(x, pw, stride=m.stride[0], pad=pad, dilation=m.dilation[0], **conv_kwargs)


class PlainConv(nn.Module):
    """Helper (not in the reference): runs an ``nn.Conv2d``'s parameters through the HIP conv kernel."""

    def __init__(self, conv: nn.Conv2d):
        super().__init__()
        self.conv = conv
        self._cache = _PackCache()

    def packed(self, dtype):
        c = self.conv
        return self._cache.get((c.weight, c.bias), dtype,
                               lambda: ops.pack_conv_weight(c.weight.data, c.bias.data if c.bias is not None else None,
                                                            dtype))


def conv_forward(conv: nn.Module, cache: _PackCache, x: ops.NHWC, **kw) -> ops.NHWC:
    """Run a plain ``nn.Conv2d`` (weights cached in packed form) or a ``SpectralNorm`` wrapper on NHWC input."""
    if isinstance(conv, SpectralNorm):
        return conv(x, **kw)
    trainable = kw.pop("trainable", False)
    if not trainable:
        _grad_guard(conv)
    if trainable and needs_grad(conv, x.t):
        pw = cache.get_plain(conv.weight, conv.bias, x.t.dtype)
        return _conv_train(x, conv.weight, conv.bias, pw, None, conv.stride[0], conv.padding[0], conv.dilation[0], kw)
    pair = isinstance(x, ops.PairMap)
    pw = cache.get((conv.weight, conv.bias) + (("pair",) if pair else ()), x.t.dtype,
                   lambda: ops.pack_conv_weight(conv.weight.data, conv.bias.data if conv.bias is not None else None,
                                                x.t.dtype, pair=pair))
    if trainable and needs_grad(conv, x.t):
        return _conv_train(x, conv.weight, conv.bias, pw, None, conv.stride[0], conv.padding[0], conv.dilation[0], kw)
    return ops.conv2d(x, pw, stride=conv.stride[0], pad=conv.padding[0], dilation=conv.dilation[0], **kw)


FUSE_BN_RESIDUAL = True      # relu(bn3(.) + residual) in one apply pass (and one backward apply pass)
# training-mode BatchNorm statistics from the producing conv's epilogue (no pass over y); CGAN_FUSE_BN_STATS=0: the
# separate statistics pass (same-box A/B)
import os as _os
FUSE_BN_STATS = _os.environ.get("CGAN_FUSE_BN_STATS", "1") != "0"
# relu(bn3(.) + skip)'s derivative in the NEXT bottleneck's first data-gradient conv instead of in bn3's backward (round 6;
# autograd.claim_relu_mask); CGAN_FUSE_RELU_MASK=0: the mask in BatchNorm's backward (same-box A/B, bit-identity test)
FUSE_RELU_MASK = _os.environ.get("CGAN_FUSE_RELU_MASK", "1") != "0"


def conv_bn_forward(conv: nn.Conv2d, bn, cache: _PackCache, x: ops.NHWC, pad_mode=ops.PAD_ZERO, pad=None, **kw) -> ops.NHWC:
    """``act(bn(conv(x)) + residual)``; ``bn`` may be None.

    * inference (no gradient wanted, BatchNorm in eval mode): the BatchNorm is folded into the conv weights (the same
      algebra as the reference's ``--fuse`` path, bn_fusion.py:121-132) and everything rides in the conv epilogue;
    * training (BatchNorm in training mode and / or a gradient wanted): conv (``autograd.ConvFn``) -> batch-statistics
      BatchNorm + residual add + activation in one apply pass (``autograd.BatchNormActFn``, running statistics
      updated like nn.BatchNorm2d), each with its HIP backward."""
    train_bn = bn is not None and bn.training
    grad = needs_grad(conv, x.t) or (bn is not None and needs_grad(bn))
    p = conv.padding[0] if pad is None else pad
    if not train_bn and not grad:
        pair = isinstance(x, ops.PairMap)       # split-precision inference (G.set_compute_dtype("pair16"))

        def build():
            if bn is None:
                return ops.pack_conv_weight(conv.weight.data, conv.bias.data if conv.bias is not None else None, x.t.dtype,
                                            pair=pair)
            w, b = ops.fold_bn(conv.weight.data, conv.bias.data if conv.bias is not None else None,
                               bn.weight.data if bn.affine else None, bn.bias.data if bn.affine else None,
                               bn.running_mean, bn.running_var, bn.eps)
            return ops.pack_conv_weight(w, b, x.t.dtype, pair=pair)

        params = [conv.weight, conv.bias]
        if bn is not None:
            params += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        if pair:
            params = params + ["pair"]
        pw = cache.get(params, x.t.dtype, build)
        return ops.conv2d(x, pw, stride=conv.stride[0], pad=p, dilation=conv.dilation[0], pad_mode=pad_mode, **kw)

    if bn is not None and not bn.training:
        raise NotImplementedError("climategan_amd: an eval-mode BatchNorm under autograd has no HIP backward (the "
                                  "reference trains with BatchNorm in training mode)")
    from .autograd import BatchNormActFn

    act, slope, residual = kw.pop("act", ops.ACT_NONE), kw.pop("slope", 0.2), kw.pop("residual", None)
    sole_consumer = kw.pop("sole_consumer", False)
    if kw.get("in_upsample") or kw.get("residual_upsample"):
        raise NotImplementedError("conv_bn_forward: folded upsamples are not used on the training path")
    pw = cache.get_plain(conv.weight, conv.bias, x.t.dtype)
    passthrough = None
    if kw.pop("passthrough", False):
        # (ResNet bottleneck) the conv also hands x through for the block's residual add: see autograd.ConvPassFn
        from .autograd import ConvPassFn

        assert bn is not None and residual is None
        pcfg = dict(c_in=x.c, stride=conv.stride[0], pad=p, dilation=conv.dilation[0], pad_mode=pad_mode, want_stats=FUSE_BN_STATS)
        if sole_consumer and FUSE_RELU_MASK:
            # x has no other reader: the derivative of the ReLU that produced it moves into this conv's data gradient
            from .autograd import claim_relu_mask
            pcfg["mask_input"] = claim_relu_mask(x.t)
        y_t, pass_t = ConvPassFn.apply(x.t, conv.weight, conv.bias, pw, pcfg)
        y, passthrough = ops.NHWC(y_t, conv.weight.shape[0]), ops.NHWC(pass_t, x.c)
        conv_stats = pcfg.get("stats_out")
    if bn is None:
        return _conv_train(x, conv.weight, conv.bias, pw, None, conv.stride[0], p, conv.dilation[0],
                           dict(act=act, slope=slope, residual=residual, pad_mode=pad_mode))
    if passthrough is None:
        conv_stats = None
        if FUSE_BN_STATS:
            y, conv_stats = _conv_train(x, conv.weight, conv.bias, pw, None, conv.stride[0], p, conv.dilation[0],
                                        dict(pad_mode=pad_mode, want_stats=True))
        else:
            y = _conv_train(x, conv.weight, conv.bias, pw, None, conv.stride[0], p, conv.dilation[0], dict(pad_mode=pad_mode))
    if residual is not None and not FUSE_BN_RESIDUAL:              # A/B switch for tools / tests: the unfused tail
        from . import functional as Fn
        out_t = BatchNormActFn.apply(y.t, bn.weight if bn.affine else None, bn.bias if bn.affine else None,
                                     bn.running_mean, bn.running_var, y.c, bn.eps,
                                     bn.momentum if bn.momentum is not None else 0.1, ops.ACT_NONE, slope,
                                     bn.num_batches_tracked if bn.track_running_stats else None)
        return Fn.add_act(ops.NHWC(out_t, y.c), residual, act, slope)
    if residual is not None and (residual.c != y.c or residual.t.shape != y.t.shape):
        raise ValueError("conv_bn_forward: residual %s does not match the conv output %s"
                         % (tuple(residual.t.shape), tuple(y.t.shape)))
    out_t = BatchNormActFn.apply(y.t, bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn.running_mean,
                                 bn.running_var, y.c, bn.eps, bn.momentum if bn.momentum is not None else 0.1, act,
                                 slope, bn.num_batches_tracked if bn.track_running_stats else None,   # counter += 1 in-kernel
                                 residual.t if residual is not None else None, conv_stats)
    return ops.NHWC(out_t, y.c) if passthrough is None else (ops.NHWC(out_t, y.c), passthrough)


class SPADE(nn.Module):
    """SPADE de-normalisation (reference climategan/norms.py:146-186), fused into one HIP kernel.

    Parameters/keys: ``mlp_shared.0.{weight,bias}``, ``mlp_gamma.{weight,bias}``, ``mlp_beta.{weight,bias}``.
    ``forward_nhwc`` takes the precomputed instance-norm statistics of x (shared between norm_0 and norm_s of
    a SPADEResnetBlock, which normalise the same tensor) and optionally applies the block's LeakyReLU.
    """

    def __init__(self, param_free_norm_type, kernel_size, norm_nc, cond_nc):
        super().__init__()
        if param_free_norm_type == "instance":
            self.param_free_norm = nn.InstanceNorm2d(norm_nc, affine=False)
        elif param_free_norm_type == "batch":
            self.param_free_norm = nn.BatchNorm2d(norm_nc, affine=False)
        else:
            raise ValueError("%s is not a recognized param-free norm type in SPADE" % param_free_norm_type)
        self.param_free_norm_type = param_free_norm_type
        nhidden = 128  # hard-coded in the reference (norms.py:163)
        pw = kernel_size // 2
        self.mlp_shared = nn.Sequential(nn.Conv2d(cond_nc, nhidden, kernel_size=kernel_size, padding=pw), nn.ReLU())
        self.mlp_gamma = nn.Conv2d(nhidden, norm_nc, kernel_size=kernel_size, padding=pw)
        self.mlp_beta = nn.Conv2d(nhidden, norm_nc, kernel_size=kernel_size, padding=pw)
        self.kernel_size = kernel_size
        self._cache = _PackCache()

    def packed(self, dtype) -> ops.PackedSpade:
        ps = (self.mlp_shared[0].weight, self.mlp_shared[0].bias, self.mlp_gamma.weight, self.mlp_gamma.bias,
              self.mlp_beta.weight, self.mlp_beta.bias)
        return self._cache.get(ps, dtype, lambda: ops.pack_spade_weights(*[p.data for p in ps], dtype))

    def forward_nhwc(self, x: ops.NHWC, cond: ops.NHWC, stats=None, act=ops.ACT_NONE, x_upsample=False) -> ops.NHWC:
        if self.kernel_size != 3:
            raise NotImplementedError("SPADE: only kernel_size 3 is supported")
        if isinstance(x, ops.PairMap):
            return self._forward_pair(x, cond, stats, act, x_upsample)
        batch_stats = False
        if self.param_free_norm_type == "batch":
            # nn.BatchNorm2d(affine=False) (norms.py:152-153): eval mode normalises with the running statistics
            bn = self.param_free_norm
            if bn.training:
                from . import autograd as _ag
                if _ag.BN_GROUPS != 1:
                    # the grouped-BatchNorm context (the trainer's merged-domain trunk) covers autograd.BatchNormActFn only;
                    # here it would silently normalise with merged-batch statistics instead of per-domain ones
                    raise NotImplementedError("SPADE with a batch param-free norm inside autograd.bn_groups(%d): run the "
                                              "SPADE mask decoder per domain (as the trainer does)" % _ag.BN_GROUPS)
                # batch statistics over (n, h, w) -- those of the up-sampled tensor when the x2 nearest upsample is
                # folded in: every pixel repeated four times gives the same mean and biased variance, only the running
                # variance's n/(n-1) factor sees the four-fold count (corrected below)
                flat = ops.NHWC(x.t.detach().view(1, x.n * x.h * x.w, 1, x.t.shape[-1]), x.c)   # SpadeFn differentiates
                mean, rstd, _, _ = ops.batchnorm_train_stats(
                    flat, None, None, bn.running_mean if bn.track_running_stats else None,
                    bn.running_var if bn.track_running_stats else None,
                    bn.num_batches_tracked if bn.track_running_stats else None, bn.eps,
                    bn.momentum if bn.momentum is not None else 0.1)
                if x_upsample and bn.track_running_stats:
                    n0 = x.n * x.h * x.w
                    mom = bn.momentum if bn.momentum is not None else 0.1
                    with torch.no_grad():
                        bn.running_var.add_(rstd[0, :x.c].pow(-2) - bn.eps,
                                            alpha=mom * (4 * n0 / (4 * n0 - 1.0) - n0 / (n0 - 1.0)))
                stats = (mean.expand(x.n, -1).contiguous(), rstd.expand(x.n, -1).contiguous())
                batch_stats = True
            else:
                _grad_guard(self)
                stats = ops.bn_eval_stats(bn, x.n)
        elif stats is None:
            stats = ops.instnorm_stats(ops.detached(x), eps=self.param_free_norm.eps)
        if needs_grad(self, x.t, cond.t):
            from .autograd import SpadeFn
            cfg = dict(c=x.c, cond_c=cond.c, act=act, slope=0.2, x_upsample=bool(x_upsample), batch_stats=batch_stats)
            y_t = SpadeFn.apply(x.t, cond.t, stats[0], stats[1], self.mlp_shared[0].weight, self.mlp_shared[0].bias,
                                self.mlp_gamma.weight, self.mlp_gamma.bias, self.mlp_beta.weight, self.mlp_beta.bias,
                                self.packed(x.t.dtype), cfg)
            return ops.NHWC(y_t, x.c)
        return ops.spade_fused(x, stats[0], stats[1], cond, self.packed(x.t.dtype), act=act, x_upsample=x_upsample)

    def _forward_pair(self, x: "ops.PairMap", cond: "ops.PairMap", stats, act, x_upsample) -> "ops.PairMap":
        """Split-precision inference (round 5; ``G.float()`` / ``set_compute_dtype("split24")``: the reference's fp32 run):
        the fused kernel multiplies a 16-bit hidden map, so here SPADE runs as the reference writes it (norms.py:174-186) --
        nearest-resized conditioning map, mlp_shared (+ ReLU), mlp_gamma, mlp_beta as split-precision convolutions, the
        de-normalisation in fp32 on the sums of the components (cgan_pair_spade_apply)."""
        _grad_guard(self)
        if not isinstance(cond, ops.PairMap):
            raise NotImplementedError("SPADE on split maps: the conditioning map must be a split map too")
        if self.param_free_norm_type == "batch":
            # nn.BatchNorm2d(affine=False) of the SPADE mask decoder (norms.py:152-153): split precision is an inference mode,
            # eval normalises with the running statistics
            if self.param_free_norm.training:
                raise NotImplementedError("SPADE on split maps: a batch param-free norm in training mode (split precision is "
                                          "an inference mode)")
            stats = ops.bn_eval_stats(self.param_free_norm, x.n)
        elif stats is None:
            stats = ops.instnorm_stats(x, eps=self.param_free_norm.eps)
        h, w = (x.h * 2, x.w * 2) if x_upsample else (x.h, x.w)
        seg = cond if (cond.h, cond.w) == (h, w) else ops.resize_nearest(cond, (h, w))
        dt = x.t.dtype

        def pk(conv):
            return self._cache.get((conv.weight, conv.bias, "pair"), dt,
                                   lambda: ops.pack_conv_weight(conv.weight.data, conv.bias.data, dt, pair=True))

        actv = ops.conv2d(seg, pk(self.mlp_shared[0]), pad=1, act=ops.ACT_RELU)
        gamma = ops.conv2d(actv, pk(self.mlp_gamma), pad=1)
        beta = ops.conv2d(actv, pk(self.mlp_beta), pad=1)
        del actv
        return ops.pair_spade_apply(x, stats[0], stats[1], gamma, beta, act=act, x_upsample=x_upsample)

    def forward(self, x, segmap, compute_dtype=None):
        """Reference signature: NCHW tensors in, NCHW fp32 out."""
        dt = compute_dtype or DEFAULT_COMPUTE_DTYPE
        from . import functional as Fn
        xs = Fn.from_nchw(x, dt)
        cond = Fn.from_nchw(segmap, dt, cs=ops.cs4(segmap.shape[1]))
        return Fn.to_nchw(self.forward_nhwc(xs, cond)).to(x.dtype)


def freeze_spectral_norm(root: nn.Module, frozen: bool = True) -> nn.Module:
    """Opt-in INFERENCE mode: every SpectralNorm under ``root`` takes its sigma from one last power iteration
    at its next forward and then keeps ``w_bar / sigma`` packed -- no power iteration, no re-pack and no u / v update per
    call any more (the Painter's 23 + the mask decoder's wrapped convs: 5 launches + 208 MB of weight reads per forward).
    The default (``frozen=False``) is the reference's behaviour: one power iteration on EVERY forward, eval included
    (norms.py:141-143), so that outputs drift with the number of calls; a frozen model is deterministic.  Training a
    frozen module raises (the backward of w_bar / sigma needs this forward's u, v)."""
    for m in root.modules():
        if isinstance(m, SpectralNorm):
            m.frozen = bool(frozen)
            m._frozen_pack = None
    return root


def spectral_norm_step_all(root: nn.Module, dtype) -> None:
    """Run this forward's power iteration + ``w_bar / sigma`` re-pack for EVERY SpectralNorm under ``root`` in
    five launches (ops.SpectralNormGroup).  Each wrapper then consumes its pre-packed weight exactly once in
    ``packed()``; a wrapper called again within the same forward falls back to its own per-layer iteration, so
    the reference's "one power iteration per call" semantics (norms.py:141-143) hold."""
    sns = [m for m in root.modules() if isinstance(m, SpectralNorm) and not m.frozen]
    if not sns:
        return
    params = [m.sn_params() for m in sns]
    grp = getattr(root, "_sn_group", None)
    if grp is None or not grp.matches(params, dtype):
        grp = ops.SpectralNormGroup(params, dtype)
        object.__setattr__(root, "_sn_group", grp)
    packed = grp.step()
    snap = None
    if torch.is_grad_enabled():
        # the backward of w_bar / sigma needs THIS forward's sigma, u, v (the next forward power-iterates them in
        # place): one multi-tensor copy per kind for the whole network instead of three clones per layer
        us = torch._foreach_mul([p[1] for p in params], 1.0)
        vs = torch._foreach_mul([p[2] for p in params], 1.0)
        snap = (grp.sigma.clone(), us, vs)
    for i, (m, pk) in enumerate(zip(sns, packed)):
        m._prepacked = pk
        m._sigma = grp.sigma[i:i + 1]
        m._sn_snapshot = None if snap is None else (snap[0][i:i + 1], snap[1][i], snap[2][i])
