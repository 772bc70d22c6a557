"""Host-side mirror of the reference's ``climategan/discriminator.py``: multi-scale spectral-norm PatchGAN for the
Painter (D["p"]) and the 5-conv ADVENT discriminators (D["m"]["Advent"], D["s"]["Advent"]).

Same class names, constructor arguments, child-module names and state-dict keys as the reference
(``p.discriminator_{i}.model{j}.0.module.{bias,weight_u,weight_v,weight_bar}``, ``{m,s}.Advent.{0,2,4,6,8}.module.*``),
forward in HIP: NHWC 16-bit activations, 4x4 stride-2 / stride-1 MFMA convs with the LeakyReLU fused into the conv
epilogue where no norm sits in between, instance-norm statistics + fused normalise/LeakyReLU otherwise, 3x3/s2
average pool between scales.  Under autograd the same modules record ``autograd.ConvFn`` / ``InstNormActFn`` nodes whose
backward is HIP as well (conv data / weight gradients, instance-norm + LeakyReLU backward, spectral-norm gradient).
"""
import functools

import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .norms import DEFAULT_COMPUTE_DTYPE, SpectralNorm, needs_grad, spectral_norm_step_all


def create_discriminator(opts, device, no_init=False, verbose=0):
    """reference discriminator.py:16-39: unless ``no_init``, ``init_weights`` per task with
    ``opts.dis[task].init_type / init_gain`` (defaults.yaml:218-219,231-232,237-238: xavier, 0.02).  Spectral-norm
    wrapped convs (the PatchGAN and, with ``gan_type: WGAN_norm``, the ADVENT nets) have no ``weight`` attribute and are
    skipped (tutils.py:58-60); with ``gan_type`` in {GAN, WGAN, WGAN_gp} the ADVENT discriminators are plain
    ``nn.Conv2d`` stacks and do get drawn.  (With ``no_init`` the reference returns the module without moving it to
    ``device``, discriminator.py:18-19; the mirror always moves it -- a CPU discriminator cannot run here.)"""
    from .tutils import init_weights

    disc = OmniDiscriminator(opts)
    if no_init:
        return disc.to(device)
    for task, model in disc.items():
        nets = list(model.items()) if isinstance(model, nn.ModuleDict) else [("", model)]
        for domain, net in nets:
            node = opts.dis[task]
            it = node["init_type"] if "init_type" in node else "xavier"
            ig = node["init_gain"] if "init_gain" in node else 0.02
            init_weights(net, init_type=it, init_gain=ig, verbose=verbose,
                         caller=("create_discriminator %s %s" % (task, domain)).strip())
    return disc.to(device)


def get_norm_layer(norm_type="instance"):
    """reference discriminator.py:64-79"""
    if not norm_type:
        norm_type = "instance"
    if norm_type == "batch":
        return functools.partial(nn.BatchNorm2d, affine=True)
    if norm_type == "instance":
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm_type == "none":
        return None
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def define_D(input_nc, ndf, n_layers=3, norm="batch", use_sigmoid=False, get_intermediate_features=False, num_D=1):
    """reference discriminator.py:42-61"""
    return MultiscaleDiscriminator(input_nc, ndf, n_layers=n_layers, norm_layer=get_norm_layer(norm_type=norm),
                                   use_sigmoid=use_sigmoid, get_intermediate_features=get_intermediate_features,
                                   num_D=num_D)


def _is_instance_norm(norm_layer):
    f = norm_layer.func if isinstance(norm_layer, functools.partial) else norm_layer
    return f == nn.InstanceNorm2d


def _to_nchw(o: ops.NHWC, module, dtype=None):
    """The reference's NCHW fp32 tensor of a feature map, with its graph under autograd (``autograd.ToNchwFn``).  The
    package's own trainer asks for ``nhwc=True`` instead and skips the 18 layout passes per call."""
    y = Fn.to_nchw(o)
    return y if dtype is None else y.to(dtype)


class NLayerDiscriminator(nn.Module):
    """PatchGAN (reference discriminator.py:82-182): model0 = SN-conv4x4s2 + LReLU; model1..n-1 = SN-conv4x4s2 +
    norm + LReLU; model_n = SN-conv4x4s1 + norm + LReLU; model_{n+1} = SN-conv4x4s1 -> 1 channel."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False,
                 get_intermediate_features=True):
        super().__init__()
        if not _is_instance_norm(norm_layer):
            raise NotImplementedError("NLayerDiscriminator: only the instance-norm configuration (defaults.yaml:217) "
                                      "has a HIP path")
        if use_sigmoid:
            raise NotImplementedError("NLayerDiscriminator: use_sigmoid=True has no HIP path (default False)")
        use_bias = True  # norm_layer == InstanceNorm2d (reference discriminator.py:93-96)
        self.get_intermediate_features = get_intermediate_features
        self.n_layers = n_layers
        self.compute_dtype = DEFAULT_COMPUTE_DTYPE
        kw, padw = 4, 1
        seq = [[SpectralNorm(nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw)), nn.LeakyReLU(0.2, True)]]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            seq += [[SpectralNorm(nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw,
                                            bias=use_bias)), norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        seq += [[SpectralNorm(nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw,
                                        bias=use_bias)), norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]]
        seq += [[SpectralNorm(nn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw))]]
        for n, mods in enumerate(seq):
            mods[0].trainable = True
            self.add_module("model" + str(n), nn.Sequential(*mods))

    def forward_nhwc(self, x: ops.NHWC):
        """list of n_layers+2 NHWC feature maps (post-activation, like the reference's ``results[1:]``)."""
        outs = []
        y = x
        last = self.n_layers + 1
        for n in range(last + 1):
            sub = getattr(self, "model" + str(n))
            has_norm = len(sub) == 3
            fuse_act = (len(sub) == 2)  # conv + LeakyReLU with no norm in between
            y = sub[0](y, act=ops.ACT_LRELU if fuse_act else ops.ACT_NONE, slope=0.2)
            if has_norm:
                if needs_grad(self, y.t):
                    from .autograd import InstNormActFn
                    y = ops.NHWC(InstNormActFn.apply(y.t, y.c, sub[1].eps, ops.ACT_LRELU, 0.2), y.c)
                else:
                    mean, rstd = ops.instnorm_stats(y, eps=sub[1].eps)
                    y = ops.norm_act_apply(y, mean, rstd, act=ops.ACT_LRELU, slope=0.2)
            outs.append(y)
        return outs

    def forward(self, input, nhwc=False):
        x = Fn.from_nchw(input, self.compute_dtype)
        outs = self.forward_nhwc(x)
        if not nhwc:
            outs = [_to_nchw(o, self) for o in outs]
        return outs if self.get_intermediate_features else outs[-1]


class MultiscaleDiscriminator(nn.Module):
    """reference discriminator.py:190-239 (pix2pixHD): ``num_D`` PatchGANs on a 3x3/s2 average-pooled pyramid."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, use_sigmoid=False,
                 get_intermediate_features=True, num_D=3):
        super().__init__()
        self.n_layers, self.ndf, self.norm_layer = n_layers, ndf, norm_layer
        self.use_sigmoid, self.get_intermediate_features, self.num_D = use_sigmoid, get_intermediate_features, num_D
        self.compute_dtype = DEFAULT_COMPUTE_DTYPE
        for i in range(num_D):
            self.add_module("discriminator_%d" % i, NLayerDiscriminator(
                input_nc=input_nc, ndf=ndf, n_layers=n_layers, norm_layer=norm_layer, use_sigmoid=use_sigmoid,
                get_intermediate_features=get_intermediate_features))
        self.downsample = nn.AvgPool2d(3, stride=2, padding=[1, 1], count_include_pad=False)

    def forward(self, input, nhwc=False):
        """``nhwc=False``: the reference's list[num_D] of list[n_layers+2] NCHW tensors (no-grad callers);
        ``nhwc=True``: the same structure of ``ops.NHWC`` maps, differentiable (training path)."""
        spectral_norm_step_all(self, self.compute_dtype)      # all 6*num_D power iterations + packs, batched
        x = Fn.from_nchw(input, self.compute_dtype)        # differentiable w.r.t. an NCHW input (autograd.FromNchwFn)
        result = []
        for i in range(self.num_D):
            D = getattr(self, "discriminator_%d" % i)
            outs = D.forward_nhwc(x)
            if not nhwc:
                outs = [_to_nchw(o, self, input.dtype if torch.is_tensor(input) else None) for o in outs]
            result.append(outs if self.get_intermediate_features else [outs[-1]])
            if i + 1 < self.num_D:
                if torch.is_grad_enabled() and x.t.requires_grad:
                    from .autograd import AvgPool3x3s2Fn
                    x = ops.NHWC(AvgPool3x3s2Fn.apply(x.t, x.c), x.c)
                else:
                    x = ops.avgpool3x3s2(x)
        return result


class FCDiscriminator(nn.Sequential):
    """ADVENT discriminator (reference ``get_fc_discriminator``, discriminator.py:327-361): 5 x conv4x4 s2 p1 with
    LeakyReLU(0.2) in between; spectral norm when ``use_norm``.  Children 0,2,4,6,8 are the convs, as in the
    reference's nn.Sequential, so the state-dict keys match."""

    def __init__(self, num_classes=2, ndf=64, use_norm=False):
        chans = [num_classes, ndf, ndf * 2, ndf * 4, ndf * 8, 1]
        mods = []
        for i in range(5):
            conv = nn.Conv2d(chans[i], chans[i + 1], kernel_size=4, stride=2, padding=1)
            mods.append(SpectralNorm(conv) if use_norm else conv)
            if i < 4:
                mods.append(nn.LeakyReLU(negative_slope=0.2, inplace=True))
        super().__init__(*mods)
        self.compute_dtype = DEFAULT_COMPUTE_DTYPE
        self._caches = {}

    def forward(self, input, nhwc=False):
        from .norms import _PackCache, conv_forward

        spectral_norm_step_all(self, self.compute_dtype)
        y = Fn.from_nchw(input, self.compute_dtype)
        for i, idx in enumerate((0, 2, 4, 6, 8)):
            cache = self._caches.setdefault(idx, _PackCache())
            if i == 0 and y.c == 2 * self._in_channels():
                y = self._first_conv_on_pair(y)
                continue
            if isinstance(self[idx], SpectralNorm):
                self[idx].trainable = True
                y = conv_forward(self[idx], cache, y, act=ops.ACT_LRELU if i < 4 else ops.ACT_NONE, slope=0.2)
            else:
                y = conv_forward(self[idx], cache, y, act=ops.ACT_LRELU if i < 4 else ops.ACT_NONE, slope=0.2,
                                 trainable=True)
        return y if nhwc else _to_nchw(y, self, input.dtype if torch.is_tensor(input) else None)


    def _in_channels(self):
        m = self[0].module if isinstance(self[0], SpectralNorm) else self[0]
        return m.in_channels

    def _first_conv_on_pair(self, y: ops.NHWC) -> ops.NHWC:
        """The first conv on a (hi | lo) pair map (losses.advent_input): conv(hi + lo) = conv([hi | lo], [w | w]).  The
        power iteration of this forward has run (spectral_norm_step_all); its sigma scales the duplicated weights."""
        from .norms import _conv_train, needs_grad

        sn = self[0] if isinstance(self[0], SpectralNorm) else None
        m = sn.module if sn is not None else self[0]
        w = getattr(m, sn.name + "_bar") if sn is not None else m.weight
        sigma = snap = None
        if sn is not None:
            sn.packed(y.t.dtype)                                 # consumes the batched step's result (or iterates)
            sigma = sn._sigma
            snap, sn._sn_snapshot = sn._sn_snapshot, None
            owned = sn._pre_used and snap is not None
            if not owned:
                snap = (sn._sigma, getattr(m, sn.name + "_u").data, getattr(m, sn.name + "_v").data)
        pw = ops.pack_conv_weight(torch.cat([w.data, w.data], 1), m.bias.data if m.bias is not None else None,
                                  y.t.dtype, sigma)
        kw = dict(act=ops.ACT_LRELU, slope=0.2)
        if needs_grad(m, y.t):
            kw.update(pair_in=True, sn_owned=bool(sn is not None and owned))
            return _conv_train(y, w, m.bias, pw, snap if sn is not None else None, m.stride[0], m.padding[0],
                               m.dilation[0], kw)
        return ops.conv2d(y, pw, stride=m.stride[0], pad=m.padding[0], dilation=m.dilation[0], **kw)


def get_fc_discriminator(num_classes=2, ndf=64, use_norm=False):
    return FCDiscriminator(num_classes, ndf, use_norm)


class OmniDiscriminator(nn.ModuleDict):
    """reference discriminator.py:242-324"""

    def __init__(self, opts):
        super().__init__()
        if "p" in opts.tasks:
            if opts.dis.p.use_local_discriminator:
                def mk():
                    return define_D(input_nc=3, ndf=opts.dis.p.ndf, n_layers=opts.dis.p.n_layers, norm=opts.dis.p.norm,
                                    use_sigmoid=opts.dis.p.use_sigmoid,
                                    get_intermediate_features=opts.dis.p.get_intermediate_features,
                                    num_D=opts.dis.p.num_D)
                self["p"] = nn.ModuleDict({"global": mk(), "local": mk()})
            else:
                self["p"] = define_D(input_nc=4,  # image + mask
                                     ndf=opts.dis.p.ndf, n_layers=opts.dis.p.n_layers, norm=opts.dis.p.norm,
                                     use_sigmoid=opts.dis.p.use_sigmoid,
                                     get_intermediate_features=opts.dis.p.get_intermediate_features,
                                     num_D=opts.dis.p.num_D)
        if "m" in opts.tasks and opts.gen.m.use_advent:
            if opts.dis.m.architecture == "base":
                self["m"] = nn.ModuleDict({"Advent": get_fc_discriminator(
                    num_classes=2, use_norm=opts.dis.m.gan_type == "WGAN_norm")})
            elif opts.dis.m.architecture == "OmniDiscriminator":
                self["m"] = nn.ModuleDict({"Advent": define_D(
                    input_nc=2, ndf=opts.dis.m.ndf, n_layers=opts.dis.m.n_layers, norm=opts.dis.m.norm,
                    use_sigmoid=opts.dis.m.use_sigmoid,
                    get_intermediate_features=opts.dis.m.get_intermediate_features, num_D=opts.dis.m.num_D)})
            else:
                raise Exception("This Discriminator is currently not supported!")
        if "s" in opts.tasks and opts.gen.s.use_advent:
            self["s"] = nn.ModuleDict({"Advent": get_fc_discriminator(
                num_classes=11, use_norm=opts.dis.s.gan_type == "WGAN_norm")})

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float16 or torch.bfloat16")
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
        return self

    # dtype casts select the kernels' 16-bit type; the fp32 master parameters stay (see OmniGenerator.half)
    def half(self):
        return self.set_compute_dtype(torch.float16)

    def bfloat16(self):
        return self.set_compute_dtype(torch.bfloat16)

    def float(self):
        return self

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype in (torch.float16, torch.bfloat16):
            self.set_compute_dtype(dtype)
        if device is not None:
            return super().to(device, non_blocking=non_blocking)
        return self
