"""DADA depth decoder -- mirror of the reference's ``climategan/depth.py`` (DADADepthDecoder; the non-default
BaseDepthDecoder is out of scope)."""
import torch.nn as nn

from . import functional as Fn
from . import ops
from .blocks import Conv2dBlock, InterpolateNearest2d
from .norms import _PackCache, conv_bn_forward
from .utils import find_target_size


def create_depth_decoder(opts, no_init=False, verbose=0):
    """reference depth.py:9-22"""
    if opts.gen.d.architecture == "base":
        raise NotImplementedError("BaseDepthDecoder (non-default) has no HIP path")
    return DADADepthDecoder(opts)


class DADADepthDecoder(nn.Module):
    """reference depth.py:25-158: enc4_1/2/3 (1x1, 3x3 reflect, 1x1; BN + LeakyReLU), optional feature-fusion output
    ``z_depth = dec4(z4_enc)``, then nearest x2 + 3x3 (128->32) + 1x1 (32->1) and the channel mean."""

    def __init__(self, opts):
        super().__init__()
        if opts.gen.encoder.architecture == "deeplabv3" and opts.gen.deeplabv3.backbone == "mobilenet":
            raise NotImplementedError("DADADepthDecoder: mobilenet backbone has no HIP path")
        res_dim, mid_dim = 2048, 512
        self.do_feat_fusion = False
        if opts.gen.m.use_dada or ("s" in opts.tasks and opts.gen.s.use_dada):
            self.do_feat_fusion = True
            self.dec4 = Conv2dBlock(128, res_dim, 1, stride=1, padding=0, bias=True, activation="lrelu", norm="none")
        self.relu = nn.ReLU(inplace=True)
        kw = dict(stride=1, bias=False, activation="lrelu", pad_type="reflect", norm="batch")
        self.enc4_1 = Conv2dBlock(res_dim, mid_dim, 1, padding=0, **kw)
        self.enc4_2 = Conv2dBlock(mid_dim, mid_dim, 3, padding=1, **kw)
        self.enc4_3 = Conv2dBlock(mid_dim, 128, 1, padding=0, **kw)
        self.upsample = None
        if opts.gen.d.upsample_featuremaps:
            self.upsample = nn.Sequential(InterpolateNearest2d(), Conv2dBlock(128, 32, 3, padding=1, **kw),
                                          nn.Conv2d(32, 1, kernel_size=1, stride=1, padding=0))
        self._target_size = find_target_size(opts, "d")
        self._cache = _PackCache()

    def set_target_size(self, size):
        self._target_size = size[:2] if isinstance(size, (list, tuple)) else (size, size)

    def forward_nhwc(self, z, passthrough=False):
        """``passthrough`` (the trainer's merged trunk, training mode): a third result, the latent handed through enc4_1's
        autograd node -- the latent's other readers (segmentation / mask decoders) take THAT tensor, so their gradients
        reach the encoder through enc4_1's data-gradient kernel (summed in its epilogue, autograd.ConvPassFn) instead of
        an element-wise sum of 2048-channel maps by the autograd engine."""
        if isinstance(z, (list, tuple)):
            z = z[0]
        z_pass = None
        if passthrough:
            z1, z_pass = self.enc4_1.forward_nhwc(z, passthrough=True)
        else:
            z1 = self.enc4_1.forward_nhwc(z)
        z4 = self.enc4_3.forward_nhwc(self.enc4_2.forward_nhwc(z1))
        z_depth = self.dec4.forward_nhwc(z4) if self.do_feat_fusion else None
        if self.upsample is None:
            raise NotImplementedError("DADADepthDecoder: upsample_featuremaps=False (channel mean over 128 maps) "
                                      "has no HIP path; the default config upsamples")
        up = Fn.upsample_nearest2x(z4)
        up = self.upsample[1].forward_nhwc(up)
        depth = conv_bn_forward(self.upsample[2], None, self._cache, up)   # 1 channel: the channel mean is the identity
        ts = self._target_size
        if depth.w != ts:                              # depth.py:143 (an int from the ctor; a tuple never compares equal)
            if not isinstance(ts, int):
                # the reference passes (ts, ts) to F.interpolate, which rejects a tuple of tuples (depth.py:151-153)
                raise TypeError("DADADepthDecoder: target size %r is not an int (reference depth.py:151-153 fails "
                                "the same way after set_target_size)" % (ts,))
            depth = Fn.resize_bicubic(depth, (384, 384))           # MiDaS inference size, depth.py:144-149
            depth = Fn.resize_nearest(depth, (ts, ts))             # depth.py:151-153 (both with their HIP adjoints)
        if passthrough:
            return depth, z_depth, z_pass
        return depth, z_depth

    def forward(self, z):
        d, zd = self.forward_nhwc(z)
        return Fn.to_nchw(d), zd
