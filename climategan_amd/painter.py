"""Host-side mirror of the reference's ``climategan/painter.py``: the SPADE Painter generator.

Same module tree and state-dict keys as ``PainterSpadeDecoder`` (reference painter.py:16-168; 229 keys at the
default config), so ``load_state_dict`` accepts reference checkpoints.  Forward runs entirely in HIP:
NHWC 16-bit activations, fused SPADE, MFMA convs, upsamples folded into the consumers.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .blocks import InterpolateNearest2d, SPADEResnetBlock
from .norms import (DEFAULT_COMPUTE_DTYPE, SpectralNorm, _grad_guard, _PackCache, conv_forward,  # noqa: F401
                    spectral_norm_step_all)


def create_painter(opts, no_init=False, verbose=0):
    """reference painter.py:10-13"""
    if verbose > 0:
        print("  - Add PainterSpadeDecoder Painter")
    return PainterSpadeDecoder(opts)


class PainterSpadeDecoder(nn.Module):
    def __init__(self, opts):
        super().__init__()
        latent_dim = opts.gen.p.latent_dim
        cond_nc = 3
        spade_n_up = opts.gen.p.spade_n_up
        sn = opts.gen.p.spade_use_spectral_norm
        pfn = opts.gen.p.spade_param_free_norm
        ks = 3

        self.z_nc = latent_dim
        self.spade_n_up = spade_n_up
        self.z_h = self.z_w = None
        self.compute_dtype = DEFAULT_COMPUTE_DTYPE

        self.fc = nn.Conv2d(3, latent_dim, 3, padding=1)
        self.head_0 = SPADEResnetBlock(self.z_nc, self.z_nc, cond_nc, sn, pfn, ks)
        self.G_middle_0 = SPADEResnetBlock(self.z_nc, self.z_nc, cond_nc, sn, pfn, ks)
        self.G_middle_1 = SPADEResnetBlock(self.z_nc, self.z_nc, cond_nc, sn, pfn, ks)
        self.up_spades = nn.Sequential(*[
            SPADEResnetBlock(self.z_nc // 2 ** i, self.z_nc // 2 ** (i + 1), cond_nc, sn, pfn, ks)
            for i in range(spade_n_up - 2)
        ])
        self.final_nc = self.z_nc // 2 ** (spade_n_up - 2)
        self.final_spade = SPADEResnetBlock(self.final_nc, self.final_nc, cond_nc, sn, pfn, ks)
        self.final_shortcut = None
        if opts.gen.p.use_final_shortcut:
            raise NotImplementedError("PainterSpadeDecoder: use_final_shortcut=True (non-default, "
                                      "defaults.yaml:155) has no HIP path")
        self.conv_img = nn.Conv2d(self.final_nc, 3, 3, padding=1)
        self.upsample = InterpolateNearest2d(scale_factor=2)
        for m in self.modules():          # every op of this network has a backward kernel: allow autograd
            if isinstance(m, SpectralNorm):
                m.trainable = True
        self._fc_cache = _PackCache()
        self._img_cache = _PackCache()
        self.pair_precision = False      # OmniGenerator.set_compute_dtype("split24" | "pair16"): split-precision inference

    def set_latent_shape(self, shape, is_input=True):
        """reference painter.py:115-136"""
        if isinstance(shape, (list, tuple, torch.Size)):
            self.z_h = shape[-2]
            self.z_w = shape[-1]
        elif isinstance(shape, int):
            self.z_h = self.z_w = shape
        else:
            raise ValueError("Unknown shape type:", shape)
        if is_input:
            self.z_h = self.z_h // (2 ** self.spade_n_up)
            self.z_w = self.z_w // (2 ** self.spade_n_up)

    def forward_nhwc(self, z, cond: ops.NHWC) -> ops.NHWC:
        """cond: NHWC (3 channels stored as 4).  Returns tanh(conv_img(...)) as NHWC (3 channels stored as 8)."""
        spectral_norm_step_all(self, cond.t.dtype)   # all 23 power iterations + w_bar/sigma packs, batched
        if z is None:
            assert self.z_h is not None and self.z_w is not None
            zin = Fn.resize_nearest(cond, (self.z_h, self.z_w), cs_out=8)       # painter.py:152
            y = conv_forward(self.fc, self._fc_cache, zin, trainable=True)
        else:
            y = z
        y = self.head_0.forward_nhwc(y, cond)
        y = self.G_middle_0.forward_nhwc(y, cond, x_upsample=True)
        y = self.G_middle_1.forward_nhwc(y, cond, x_upsample=True)
        for up in self.up_spades:
            y = up.forward_nhwc(y, cond, x_upsample=True)
        y = self.final_spade.forward_nhwc(y, cond, post_act="lrelu")             # painter.py:165-166
        return conv_forward(self.conv_img, self._img_cache, y, act=ops.ACT_TANH, trainable=True)  # painter.py:166-167

    def forward(self, z, cond):
        """Reference signature (painter.py:149): z None or [B,latent,z_h,z_w]; cond [B,3,H,W] NCHW."""
        dt = self.compute_dtype
        if self.pair_precision and not (torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters())):
            # split-precision inference (round 5): conditioning map and latent as split maps, every conv a split-precision
            # conv, SPADE unfused with its de-normalisation in fp32 (norms.SPADE._forward_pair)
            c = ops.pair_from_nchw(cond.float(), dt)
            zz = ops.pair_from_nchw(z.float(), dt) if z is not None else None
            return ops.nhwc_to_nchw(self.forward_nhwc(zz, c)).to(cond.dtype)
        c = Fn.from_nchw(cond, dt, cs=4)
        zz = Fn.from_nchw(z, dt) if z is not None else None
        return Fn.to_nchw(self.forward_nhwc(zz, c)).to(cond.dtype)
