"""Mask decoder -- mirror of the reference's ``climategan/masker.py`` (MaskBaseDecoder; MaskSpadeDecoder, the
non-default SPADE-conditioned variant with batch-norm SPADE, is not built yet)."""
from . import ops
from .blocks import BaseDecoder


def create_mask_decoder(opts, no_init=False, verbose=0):
    """reference masker.py:13-22"""
    if opts.gen.m.use_spade:
        raise NotImplementedError("MaskSpadeDecoder (gen.m.use_spade) has no HIP path yet")
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
        return ops.nhwc_to_nchw(self.forward_nhwc(z, cond, z_depth))
