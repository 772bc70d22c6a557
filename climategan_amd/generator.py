"""Host-side mirror of the reference's ``climategan/generator.py`` (the model API boundary).

Built so far: the Painter branch -- ``create_generator``, ``OmniGenerator.painter``, ``paint``,
``sample_painter_z`` -- with the masking ``x * (1 - m)`` and the paste ``x * (1 - m) + fake * m`` folded
into the NCHW<->NHWC edge kernels.  The Masker branch (encoder / depth / seg / mask decoders) raises
NotImplementedError until its kernels land.
"""
import torch
import torch.nn as nn

from . import ops
from .painter import create_painter


def create_generator(opts, device="cpu", latent_shape=None, no_init=False, verbose=0):
    """reference generator.py:24-61.  The reference never re-initialises the Painter (it keeps torch's default
    conv init, generator.py:30-58 only loops over ``G.decoders``), so there is nothing to init here."""
    G = OmniGenerator(opts, latent_shape, verbose, no_init)
    return G.to(device)


class OmniGenerator(nn.Module):
    def __init__(self, opts, latent_shape=None, verbose=0, no_init=False):
        super().__init__()
        self.opts = opts
        self.verbose = verbose
        self.encoder = None
        if any(t in opts.tasks for t in "msd"):
            if verbose > 0:
                print("  - Masker (encoder + d/s/m decoders): HIP path not built yet; skipped")
        self.decoders = nn.ModuleDict({})
        self.painter = nn.Module()
        if "p" in opts.tasks:
            self.painter = create_painter(opts, no_init, verbose)

    def set_compute_dtype(self, dtype):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float16 or torch.bfloat16")
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
        return self

    def encode(self, x):
        raise NotImplementedError("OmniGenerator.encode: the Masker's HIP path is not built yet")

    def decode(self, x=None, z=None, return_z=False, return_z_depth=False):
        raise NotImplementedError("OmniGenerator.decode: the Masker's HIP path is not built yet")

    def mask(self, x=None, z=None, cond=None, z_depth=None, sigmoid=True):
        raise NotImplementedError("OmniGenerator.mask: the Masker's HIP path is not built yet")

    def sample_painter_z(self, batch_size, device, force_half=False):
        """reference generator.py:179-194"""
        if self.opts.gen.p.no_z:
            return None
        z = torch.empty(batch_size, self.opts.gen.p.latent_dim, self.painter.z_h, self.painter.z_w,
                        device=device).normal_(mean=0, std=1.0)
        if force_half:
            z = z.half()
        return z

    def paint(self, m, x, no_paste=False):
        """reference generator.py:279-297: fake = painter(z, x * (1 - m)); returns x * (1 - m) + fake * m.

        m [B,1,H,W] (1 where water is painted), x [B,3,H,W] in [-1,1]; NCHW in, NCHW out (x's dtype)."""
        p = self.painter
        dt = p.compute_dtype
        z = self.sample_painter_z(x.shape[0], x.device)
        m = m.to(x.dtype)
        cond = ops.nchw_to_nhwc(x, dt, cs=4, mask=m)                     # x * (1 - m)
        zz = ops.nchw_to_nhwc(z, dt) if z is not None else None
        fake = p.forward_nhwc(zz, cond)
        if self.opts.gen.p.paste_original_content and not no_paste:
            if tuple(fake.t.shape[1:3]) != tuple(x.shape[-2:]):
                raise RuntimeError("paint: painter output %s does not match x %s (input must be a multiple of %d)"
                                   % (tuple(fake.t.shape[1:3]), tuple(x.shape[-2:]), 2 ** p.spade_n_up))
            return ops.nhwc_to_nchw(fake, paste_x=x, paste_m=m).to(x.dtype)
        return ops.nhwc_to_nchw(fake).to(x.dtype)
