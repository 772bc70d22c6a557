"""Mask decoders -- mirror of the reference's ``climategan/masker.py``: MaskBaseDecoder (default) and MaskSpadeDecoder
(``gen.m.use_spade``: the paper's final masker, SPADE blocks conditioned on depth / segmentation / image)."""
import torch.nn as nn

from . import functional as Fn
from . import ops
from .blocks import BaseDecoder, Conv2dBlock, InterpolateNearest2d, SPADEResnetBlock
from .norms import SpectralNorm


def create_mask_decoder(opts, no_init=False, verbose=0):
    """reference masker.py:13-22"""
    if opts.gen.m.use_spade:
        return MaskSpadeDecoder(opts)
    return MaskBaseDecoder(opts)


class MaskBaseDecoder(BaseDecoder):
    """reference masker.py:25-56"""

    def __init__(self, opts):
        use_v3 = opts.gen.encoder.architecture == "deeplabv3"
        if use_v3 and opts.gen.deeplabv3.backbone == "mobilenet":
            raise NotImplementedError("MaskBaseDecoder: mobilenet backbone has no HIP path")
        low = 256 if (use_v3 and opts.gen.m.use_low_level_feats) else -1
        super().__init__(n_upsample=opts.gen.m.n_upsample, n_res=opts.gen.m.n_res, input_dim=2048,
                         proj_dim=opts.gen.m.proj_dim, output_dim=opts.gen.m.output_dim, norm=opts.gen.m.norm,
                         activ=opts.gen.m.activ, pad_type=opts.gen.m.pad_type, output_activ="none",
                         low_level_feats_dim=low, use_dada=("d" in opts.tasks) and opts.gen.m.use_dada)

    def forward(self, z, cond=None, z_depth=None):
        """reference signature (blocks.py:292-313): the mask LOGITS as an NCHW fp32 tensor (with its graph under autograd).
        Every call is one forward of the spectral-norm convs: one power iteration each (norms.py:141-143)."""
        from .norms import spectral_norm_step_all
        zz = z[0] if isinstance(z, (tuple, list)) else z
        spectral_norm_step_all(self, zz.t.dtype)
        return Fn.to_nchw(self.forward_nhwc(z, cond, z_depth))


class MaskSpadeDecoder(nn.Module):
    """reference masker.py:59-231 (resnet deeplabv3 backbone): projection convs (spectral norm + BatchNorm, reflect
    padding), ``num_layers`` SPADE ResNet blocks with a batch param-free norm, each followed by a x2 nearest upsample
    (folded into the next consumer), then a spectral-norm 3x3 conv to one channel.  Eval mode normalises with the
    running statistics; training mode with batch statistics (HIP backward, including the gradient of the conditioning
    map when ``gen.m.spade.detach`` is false)."""

    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        sp = opts.gen.m.spade
        self.num_layers = sp.num_layers
        self.z_nc = sp.latent_dim
        act = "lrelu" if sp.activations.all_lrelu else None
        if opts.gen.encoder.architecture != "deeplabv3" or opts.gen.deeplabv3.backbone != "resnet":
            raise NotImplementedError("MaskSpadeDecoder: only the deeplabv3 / resnet encoder has a HIP path")
        self.input_dim = [2048, 256]
        kw = dict(padding=1, activation="lrelu", pad_type="reflect", norm="spectral_batch")
        if opts.gen.m.use_proj:
            proj = opts.gen.m.proj_dim
            self.low_level_conv = Conv2dBlock(self.input_dim[1], proj, 3, **kw)
            self.high_level_conv = Conv2dBlock(self.input_dim[0], proj, 3, **kw)
            self.merge_feats_conv = Conv2dBlock(proj * 2, self.z_nc, 3, **kw)
        else:
            self.low_level_conv = Conv2dBlock(self.input_dim[1], self.input_dim[0], 3, **kw)
            self.merge_feats_conv = Conv2dBlock(self.input_dim[0] * 2, self.z_nc, 3, **kw)
        self.spade_blocks = nn.Sequential(*[
            SPADEResnetBlock(int(self.z_nc / 2 ** i), int(self.z_nc / 2 ** (i + 1)), sp.cond_nc,
                             sp.spade_use_spectral_norm, sp.spade_param_free_norm, 3, act)
            for i in range(self.num_layers)])
        self.final_nc = int(self.z_nc / 2 ** self.num_layers)
        self.mask_conv = Conv2dBlock(self.final_nc, 1, 3, padding=1, activation="none", pad_type="reflect",
                                     norm="spectral")
        self.upsample = InterpolateNearest2d(scale_factor=2)
        for m in self.modules():          # every op of this decoder has a backward kernel: allow autograd
            if isinstance(m, SpectralNorm):
                m.trainable = True

    def forward_nhwc(self, z, cond: ops.NHWC, z_depth=None) -> ops.NHWC:
        if not isinstance(z, (list, tuple)):
            raise NotImplementedError("MaskSpadeDecoder: the deeplabv2 single-tensor latent has no HIP path")
        z_h, z_l = z
        z_l = self.low_level_conv.forward_nhwc(z_l)
        from . import functional as Fn                    # grad-aware: HIP backward when the latent carries a graph
        z_l = Fn.resize_bilinear(z_l, (z_h.h, z_h.w), align_corners=False)              # masker.py:217,221
        if self.opts.gen.m.use_proj:
            z_h = self.high_level_conv.forward_nhwc(z_h)
        y = self.merge_feats_conv.forward_nhwc(Fn.concat_channels([z_h, z_l]))          # masker.py:222-223
        for i in range(self.num_layers):
            y = self.spade_blocks[i].forward_nhwc(y, cond, x_upsample=(i > 0))          # upsample folded: :227-229
        # the last upsample, read through the conv (reflect padding on the up-sampled extent)
        c = self.mask_conv
        if y.t.requires_grad:
            # training: the weight-gradient kernel reads x either through the upsample or through the reflection, not
            # both -- materialise the (16-channel) up-sampled map
            return c.conv(Fn.upsample_nearest2x(y), pad=c.padding, pad_mode=ops.PAD_REFLECT)
        return c.conv(y, pad=c.padding, pad_mode=ops.PAD_REFLECT, in_upsample=True)

    def forward(self, z, cond, z_depth=None):
        from . import functional as Fn
        from .norms import spectral_norm_step_all
        spectral_norm_step_all(self, z[0].t.dtype)
        return Fn.to_nchw(self.forward_nhwc(z, cond, z_depth))
