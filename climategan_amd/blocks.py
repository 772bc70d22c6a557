"""Host-side mirror of the reference's ``climategan/blocks.py`` for the Painter path.

``SPADEResnetBlock`` and ``InterpolateNearest2d`` keep the reference's names, constructor arguments and
state-dict keys; the arithmetic runs in HIP.  The x2 nearest upsample that precedes most blocks in the
Painter (painter.py:154-161) is never materialised: the block reads its input through index math
(``x_upsample=True``), and instance-norm statistics are taken on the pre-upsample tensor (nearest x2
replicates every pixel 4 times, so mean and biased variance are unchanged).
"""
import torch
import torch.nn as nn

from . import ops
from .norms import DEFAULT_COMPUTE_DTYPE, SPADE, SpectralNorm, _PackCache, conv_forward


class InterpolateNearest2d(nn.Module):
    """Nearest x``scale_factor`` upsample (reference climategan/blocks.py:11-43)."""

    def __init__(self, scale_factor=2):
        super().__init__()
        self.scale_factor = scale_factor

    def forward(self, x):
        if isinstance(x, ops.NHWC):
            return ops.resize_nearest(x, (x.h * self.scale_factor, x.w * self.scale_factor))
        dt = DEFAULT_COMPUTE_DTYPE
        y = ops.resize_nearest(ops.nchw_to_nhwc(x, dt), (x.shape[-2] * self.scale_factor, x.shape[-1] * self.scale_factor))
        return ops.nhwc_to_nchw(y).to(x.dtype)


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
        stats = ops.instnorm_stats(x, eps=self.norm_0.param_free_norm.eps)
        if self.learned_shortcut:
            s = self.norm_s.forward_nhwc(x, cond, stats, act=ops.ACT_NONE, x_upsample=x_upsample)
            x_s = conv_forward(self.conv_s, self._caches["conv_s"], s)
            res, res_ups = x_s, False
        else:
            res, res_ups = x, x_upsample
        dx = self.norm_0.forward_nhwc(x, cond, stats, act=ops.ACT_LRELU, x_upsample=x_upsample)
        dx = conv_forward(self.conv_0, self._caches["conv_0"], dx)
        dx = self.norm_1.forward_nhwc(dx, cond, None, act=ops.ACT_LRELU)
        act = ops.ACT_LRELU if (self.last_activation == "lrelu" or post_act == "lrelu") else ops.ACT_NONE
        if self.last_activation == "lrelu" and post_act == "lrelu":
            raise NotImplementedError("SPADEResnetBlock: last_activation and post_act cannot both be lrelu")
        return conv_forward(self.conv_1, self._caches["conv_1"], dx, residual=res, residual_upsample=res_ups, act=act)

    def forward(self, x, seg, compute_dtype=None):
        """Reference signature: NCHW tensors in, NCHW out."""
        dt = compute_dtype or DEFAULT_COMPUTE_DTYPE
        xs = ops.nchw_to_nhwc(x, dt)
        cond = ops.nchw_to_nhwc(seg, dt, cs=ops.cs4(seg.shape[1]))
        return ops.nhwc_to_nchw(self.forward_nhwc(xs, cond)).to(x.dtype)

    def shortcut(self, x, seg):
        raise NotImplementedError("SPADEResnetBlock.shortcut is fused into forward() in this build")

    def activation(self, x):
        return torch.nn.functional.leaky_relu(x, 2e-1)
