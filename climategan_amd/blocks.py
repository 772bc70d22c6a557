"""Host-side mirror of the reference's ``climategan/blocks.py`` for the Painter path.

``SPADEResnetBlock`` and ``InterpolateNearest2d`` keep the reference's names, constructor arguments and
state-dict keys; the arithmetic runs in HIP.  The x2 nearest upsample that precedes most blocks in the
Painter (painter.py:154-161) is never materialised: the block reads its input through index math
(``x_upsample=True``), and instance-norm statistics are taken on the pre-upsample tensor (nearest x2
replicates every pixel 4 times, so mean and biased variance are unchanged).
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .norms import (DEFAULT_COMPUTE_DTYPE, SPADE, SpectralNorm, _PackCache, conv_bn_forward,
                    conv_forward)


class InterpolateNearest2d(nn.Module):
    """Nearest x``scale_factor`` upsample (reference climategan/blocks.py:11-43)."""

    def __init__(self, scale_factor=2):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        if isinstance(x, ops.PairMap):
            return ops.resize_nearest(x, (x.h * self.scale_factor, x.w * self.scale_factor))
        if isinstance(x, ops.NHWC):
            if self.scale_factor == 2:
                return Fn.upsample_nearest2x(x)
            return ops.resize_nearest(x, (x.h * self.scale_factor, x.w * self.scale_factor))
        dt = DEFAULT_COMPUTE_DTYPE
        y = Fn.resize_nearest(Fn.from_nchw(x, dt), (x.shape[-2] * self.scale_factor, x.shape[-1] * self.scale_factor))
        return Fn.to_nchw(y).to(x.dtype)


_ACTS = {"relu": ops.ACT_RELU, "lrelu": ops.ACT_LRELU, "tanh": ops.ACT_TANH, "sigmoid": ops.ACT_SIGMOID,
         "none": ops.ACT_NONE}


class Conv2dBlock(nn.Module):
    """pad -> conv (optionally spectral-norm wrapped) -> norm -> activation (reference climategan/blocks.py:49-146).

    HIP forward: the explicit pad module becomes the conv kernel's padding mode (zero / reflect); an eval-mode
    BatchNorm is folded into the weights; bias, activation and an optional residual ride in the conv epilogue."""

    def __init__(self, input_dim, output_dim, kernel_size, stride=1, padding=0, dilation=1, norm="none",
                 activation="relu", pad_type="zero", bias=True):
        super().__init__()
        self.use_bias = bias
        if pad_type == "reflect":
            self.pad = nn.ReflectionPad2d(padding)
        elif pad_type == "replicate":
            self.pad = nn.ReplicationPad2d(padding)
        elif pad_type == "zero":
            self.pad = nn.ZeroPad2d(padding)
        else:
            assert 0, "Unsupported padding type: {}".format(pad_type)
        self.pad_type, self.padding = pad_type, padding
        use_spectral_norm = False
        if norm.startswith("spectral_"):
            norm = norm.replace("spectral_", "")
            use_spectral_norm = True
        if norm == "batch":
            self.norm = nn.BatchNorm2d(output_dim)
        elif norm == "instance":
            self.norm = nn.InstanceNorm2d(output_dim)
        elif norm in ("layer", "adain"):
            raise NotImplementedError("Conv2dBlock: norm '%s' is a dead option in the default configs; no HIP path" % norm)
        elif norm == "spectral" or norm == "none":
            self.norm = None
        else:
            raise ValueError("Unsupported normalization: {}".format(norm))
        if activation in ("prelu", "selu"):
            raise NotImplementedError("Conv2dBlock: activation '%s' has no HIP path" % activation)
        if activation not in _ACTS:
            raise ValueError("Unsupported activation: {}".format(activation))
        self.activation_name = activation
        self.activation = {"relu": nn.ReLU(inplace=False), "lrelu": nn.LeakyReLU(0.2, inplace=False), "tanh": nn.Tanh(),
                           "sigmoid": nn.Sigmoid(), "none": None}[activation]
        if norm == "spectral" or use_spectral_norm:
            self.conv = SpectralNorm(nn.Conv2d(input_dim, output_dim, kernel_size, stride, dilation=dilation,
                                               bias=self.use_bias))
            self.conv.trainable = True          # conv (zero / reflect pad), residual and activation all have backward kernels
        else:
            self.conv = nn.Conv2d(input_dim, output_dim, kernel_size, stride, dilation=dilation,
                                  bias=self.use_bias if norm != "batch" else False)
        self._cache = _PackCache()

    def forward_nhwc(self, x: ops.NHWC, residual=None, passthrough=False) -> ops.NHWC:
        """``passthrough`` (training, plain conv + BatchNorm blocks only): also return x as an output of this conv's autograd
        node, for the next consumer of the same tensor (norms.conv_bn_forward / autograd.ConvPassFn)."""
        if self.pad_type == "replicate":
            raise NotImplementedError("Conv2dBlock: replicate padding has no HIP path")
        if passthrough:
            if (isinstance(self.conv, SpectralNorm) or not isinstance(self.norm, nn.BatchNorm2d) or residual is not None
                    or not self.norm.training):
                raise NotImplementedError("Conv2dBlock: passthrough is for a plain conv + training-mode BatchNorm block")
            pad_mode = ops.PAD_REFLECT if (self.pad_type == "reflect" and self.padding > 0) else ops.PAD_ZERO
            return conv_bn_forward(self.conv, self.norm, self._cache, x, pad_mode=pad_mode, pad=self.padding,
                                   act=_ACTS[self.activation_name], slope=0.2, passthrough=True)
        pad_mode = ops.PAD_REFLECT if (self.pad_type == "reflect" and self.padding > 0) else ops.PAD_ZERO
        act = _ACTS[self.activation_name]
        sn = isinstance(self.conv, SpectralNorm)
        if isinstance(self.norm, nn.InstanceNorm2d):
            if residual is not None:
                raise NotImplementedError("Conv2dBlock: residual + instance norm cannot be fused")
            y = (self.conv(x, pad=self.padding, pad_mode=pad_mode) if sn else
                 conv_bn_forward(self.conv, None, self._cache, x, pad_mode=pad_mode, pad=self.padding))
            mean, rstd = ops.instnorm_stats(y, eps=self.norm.eps)
            return ops.norm_act_apply(y, mean, rstd, act=act, slope=0.2)
        if sn:
            if self.norm is not None:                                   # spectral_batch: SN conv -> BatchNorm -> act
                if residual is not None:
                    raise NotImplementedError("Conv2dBlock: residual + spectral_batch cannot be fused")
                y = self.conv(x, pad=self.padding, pad_mode=pad_mode)
                if self.norm.training:                                  # batch statistics, HIP backward
                    from .autograd import BatchNormActFn
                    bn = self.norm
                    out_t = BatchNormActFn.apply(y.t, bn.weight if bn.affine else None, bn.bias if bn.affine else None,
                                                 bn.running_mean, bn.running_var, y.c, bn.eps,
                                                 bn.momentum if bn.momentum is not None else 0.1, act, 0.2,
                                                 bn.num_batches_tracked if bn.track_running_stats else None)
                    return ops.NHWC(out_t, y.c)
                if y.t.requires_grad:
                    raise NotImplementedError("climategan_amd: an eval-mode BatchNorm under autograd has no HIP "
                                              "backward (the reference trains with BatchNorm in training mode)")
                mean, rstd = ops.bn_eval_stats(self.norm, y.n)
                return ops.norm_act_apply(y, mean, rstd, act=act, slope=0.2)
            return self.conv(x, pad=self.padding, pad_mode=pad_mode, act=act, slope=0.2, residual=residual)
        return conv_bn_forward(self.conv, self.norm, self._cache, x, pad_mode=pad_mode, pad=self.padding, act=act,
                               slope=0.2, residual=residual)

    def forward(self, x, compute_dtype=None):
        dt = compute_dtype or DEFAULT_COMPUTE_DTYPE
        return Fn.to_nchw(self.forward_nhwc(Fn.from_nchw(x, dt))).to(x.dtype)


class ResBlock(nn.Module):
    """reference climategan/blocks.py:174-201: out = conv_b(conv_a(x)) + x (the add is fused into conv_b)."""

    def __init__(self, dim, norm="in", activation="relu", pad_type="zero"):
        super().__init__()
        self.dim, self.norm, self.activation = dim, norm, activation
        self.model = nn.Sequential(
            Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation=activation, pad_type=pad_type),
            Conv2dBlock(dim, dim, 3, 1, 1, norm=norm, activation="none", pad_type=pad_type))

    def forward_nhwc(self, x):
        return self.model[1].forward_nhwc(self.model[0].forward_nhwc(x), residual=x)


class ResBlocks(nn.Module):
    """reference climategan/blocks.py:153-171"""

    def __init__(self, num_blocks, dim, norm="in", activation="relu", pad_type="zero"):
        super().__init__()
        self.model = nn.Sequential(*[ResBlock(dim, norm=norm, activation=activation, pad_type=pad_type)
                                     for _ in range(num_blocks)])

    def forward_nhwc(self, x):
        for b in self.model:
            x = b.forward_nhwc(x)
        return x


class BaseDecoder(nn.Module):
    """reference climategan/blocks.py:206-313 (mask decoder base)."""

    def __init__(self, n_upsample=4, n_res=4, input_dim=2048, proj_dim=64, output_dim=3, norm="batch", activ="relu",
                 pad_type="zero", output_activ="tanh", low_level_feats_dim=-1, use_dada=False):
        super().__init__()
        self.low_level_feats_dim = low_level_feats_dim
        self.use_dada = use_dada
        if proj_dim != -1:
            self.proj_conv = Conv2dBlock(input_dim, proj_dim, 1, 1, 0, norm=norm, activation=activ)
        else:
            self.proj_conv = None
            proj_dim = input_dim
        if low_level_feats_dim > 0:
            self.low_level_conv = Conv2dBlock(input_dim=low_level_feats_dim, output_dim=proj_dim, kernel_size=3, stride=1,
                                              padding=1, pad_type=pad_type, norm=norm, activation=activ)
            self.merge_feats_conv = Conv2dBlock(input_dim=2 * proj_dim, output_dim=proj_dim, kernel_size=1, stride=1,
                                                padding=0, pad_type=pad_type, norm=norm, activation=activ)
        else:
            self.low_level_conv = None
        model = [ResBlocks(n_res, proj_dim, norm, activ, pad_type=pad_type)]
        dim = proj_dim
        for _ in range(n_upsample):
            model += [InterpolateNearest2d(scale_factor=2),
                      Conv2dBlock(input_dim=dim, output_dim=dim // 2, kernel_size=3, stride=1, padding=1,
                                  pad_type=pad_type, norm=norm, activation=activ)]
            dim //= 2
        model += [Conv2dBlock(input_dim=dim, output_dim=output_dim, kernel_size=3, stride=1, padding=1,
                              pad_type=pad_type, norm="none", activation=output_activ)]
        self.model = nn.Sequential(*model)

    def forward_nhwc(self, z, cond=None, z_depth=None) -> ops.NHWC:
        low = None
        if isinstance(z, (list, tuple)):
            if self.low_level_conv is None:
                z = z[0]
            else:
                z, low = z
                low = self.low_level_conv.forward_nhwc(low)
                low = Fn.resize_bilinear(low, (z.h, z.w), align_corners=False)      # blocks.py:300-302
        if z_depth is not None and self.use_dada:
            z = Fn.mul(z, z_depth)
        if self.proj_conv is not None:
            z = self.proj_conv.forward_nhwc(z)
        if low is not None:
            z = self.merge_feats_conv.forward_nhwc(Fn.concat_channels([low, z]))
        for m in self.model:
            z = m(z) if isinstance(m, InterpolateNearest2d) else m.forward_nhwc(z)
        return z


class SPADEResnetBlock(nn.Module):
    """SPADE ResNet block (reference climategan/blocks.py:325-395).

    out = shortcut(x) + conv_1(lrelu(SPADE_1(conv_0(lrelu(SPADE_0(x))))))   [+ optional last lrelu]
    with a learned 1x1 shortcut through SPADE_s when fin != fout.
    """

    def __init__(self, fin, fout, cond_nc, spade_use_spectral_norm, spade_param_free_norm, spade_kernel_size,
                 last_activation=None):
        super().__init__()
        self.fin = fin
        self.fout = fout
        self.use_spectral_norm = spade_use_spectral_norm
        self.param_free_norm = spade_param_free_norm
        self.kernel_size = spade_kernel_size
        self.learned_shortcut = fin != fout
        self.last_activation = last_activation
        fmiddle = min(fin, fout)

        self.conv_0 = nn.Conv2d(fin, fmiddle, kernel_size=3, padding=1)
        self.conv_1 = nn.Conv2d(fmiddle, fout, kernel_size=3, padding=1)
        if self.learned_shortcut:
            self.conv_s = nn.Conv2d(fin, fout, kernel_size=1, bias=False)
        if spade_use_spectral_norm:
            self.conv_0 = SpectralNorm(self.conv_0)
            self.conv_1 = SpectralNorm(self.conv_1)
            if self.learned_shortcut:
                self.conv_s = SpectralNorm(self.conv_s)

        self.norm_0 = SPADE(spade_param_free_norm, spade_kernel_size, fin, cond_nc)
        self.norm_1 = SPADE(spade_param_free_norm, spade_kernel_size, fmiddle, cond_nc)
        if self.learned_shortcut:
            self.norm_s = SPADE(spade_param_free_norm, spade_kernel_size, fin, cond_nc)
        self._caches = {k: _PackCache() for k in ("conv_0", "conv_1", "conv_s")}

    def forward_nhwc(self, x: ops.NHWC, cond: ops.NHWC, x_upsample=False, post_act=None) -> ops.NHWC:
        """x: block input (stored pre-upsample when ``x_upsample``); cond: full-resolution conditioning.
        ``post_act``: extra activation folded into the residual epilogue (the Painter applies LeakyReLU to
        final_spade's output before conv_img, painter.py:166)."""
        if self.last_activation not in (None, "lrelu"):
            raise NotImplementedError(
                "The type of activation is not supported: {}".format(self.last_activation))
        # instance-norm statistics are shared by norm_0 and norm_s (same x); a batch param-free norm brings its own
        # (computed outside the graph: SpadeFn's backward carries the statistics' dependence on x itself)
        stats = (None if self.param_free_norm == "batch" else
                 ops.instnorm_stats(ops.detached(x), eps=self.norm_0.param_free_norm.eps))
        if self.learned_shortcut:
            s = self.norm_s.forward_nhwc(x, cond, stats, act=ops.ACT_NONE, x_upsample=x_upsample)
            x_s = conv_forward(self.conv_s, self._caches["conv_s"], s, **self._tr(self.conv_s))
            res, res_ups = x_s, False
        else:
            res, res_ups = x, x_upsample
        dx = self.norm_0.forward_nhwc(x, cond, stats, act=ops.ACT_LRELU, x_upsample=x_upsample)
        dx = conv_forward(self.conv_0, self._caches["conv_0"], dx, **self._tr(self.conv_0))
        dx = self.norm_1.forward_nhwc(dx, cond, None, act=ops.ACT_LRELU)
        act = ops.ACT_LRELU if (self.last_activation == "lrelu" or post_act == "lrelu") else ops.ACT_NONE
        if self.last_activation == "lrelu" and post_act == "lrelu":
            raise NotImplementedError("SPADEResnetBlock: last_activation and post_act cannot both be lrelu")
        return conv_forward(self.conv_1, self._caches["conv_1"], dx, residual=res, residual_upsample=res_ups, act=act,
                            **self._tr(self.conv_1))

    @staticmethod
    def _tr(conv):
        """plain (non spectral-norm) convs take the training path through conv_forward's ``trainable`` switch"""
        return {} if isinstance(conv, SpectralNorm) else {"trainable": True}

    def forward(self, x, seg, compute_dtype=None):
        """Reference signature: NCHW tensors in, NCHW out."""
        dt = compute_dtype or DEFAULT_COMPUTE_DTYPE
        xs = Fn.from_nchw(x, dt)
        cond = Fn.from_nchw(seg, dt, cs=ops.cs4(seg.shape[1]))
        return Fn.to_nchw(self.forward_nhwc(xs, cond)).to(x.dtype)

    def shortcut_nhwc(self, x: ops.NHWC, cond: ops.NHWC, x_upsample=False) -> ops.NHWC:
        """The block's skip path on its own: conv_s(SPADE_s(x, seg)) when fin != fout, else x (at the conditioning map's
        resolution: a stored pre-upsample x is expanded)."""
        if not self.learned_shortcut:
            return Fn.resize_nearest(x, (cond.h, cond.w)) if x_upsample else x
        stats = (None if self.param_free_norm == "batch" else
                 ops.instnorm_stats(ops.detached(x), eps=self.norm_s.param_free_norm.eps))
        s = self.norm_s.forward_nhwc(x, cond, stats, act=ops.ACT_NONE, x_upsample=x_upsample)
        return conv_forward(self.conv_s, self._caches["conv_s"], s, **self._tr(self.conv_s))

    def shortcut(self, x, seg, compute_dtype=None):
        """Reference signature (climategan/blocks.py:387-392): NCHW tensors in, NCHW out.  ``forward`` does not call it (the
        skip path rides in conv_1's residual epilogue there); it is the same fused-SPADE + 1x1 kernels run on their own."""
        dt = compute_dtype or DEFAULT_COMPUTE_DTYPE
        xs = Fn.from_nchw(x, dt)
        cond = Fn.from_nchw(seg, dt, cs=ops.cs4(seg.shape[1]))
        return Fn.to_nchw(self.shortcut_nhwc(xs, cond)).to(x.dtype)

    def activation(self, x):
        return torch.nn.functional.leaky_relu(x, 2e-1)
