"""Host-side mirror of the reference's ``climategan/generator.py`` (the model API boundary).

``create_generator``, ``OmniGenerator.{encoder, decoders, painter}``, ``encode`` / ``decode`` / ``mask`` / ``depth`` /
``make_m_cond`` / ``paint`` / ``paint_cloudy`` / ``sample_painter_z`` / ``load_val_painter``; the masking ``x * (1 - m)``
and the paste ``x * (1 - m) + fake * m`` are folded into the NCHW<->NHWC edge kernels.  Under autograd the NHWC
variants (``paint_nhwc``, ``mask_nhwc``, ``decoders[t].forward_nhwc``) are the ones with HIP backward passes.
"""
import torch
import torch.nn as nn

from . import functional as Fn
from . import ops
from .deeplab import create_encoder, create_segmentation_decoder
from .depth import create_depth_decoder
from .masker import create_mask_decoder
from .norms import DEFAULT_COMPUTE_DTYPE, spectral_norm_step_all
from .painter import create_painter
from .tutils import init_weights


def create_generator(opts, device="cpu", latent_shape=None, no_init=False, verbose=0):
    """reference generator.py:24-61.  Unless ``no_init``: ``init_weights`` with ``opts.gen[task].init_type / init_gain``
    (defaults.yaml:92-93: xavier, 0.02) on every decoder EXCEPT the segmentation head "s" (which initialises itself,
    deeplab_v3.py:178-190) -- i.e. the depth decoder's convs and BatchNorms, the mask decoder's plain output conv and,
    for the SPADE mask decoder, its ``mlp_shared / mlp_gamma / mlp_beta`` convs and BatchNorms (spectral-norm wrapped
    convs have no ``weight`` and are skipped, tutils.py:58-60); the encoder only when its architecture is "base".  The
    Painter is never re-initialised by the reference (it keeps torch's default conv init)."""
    G = OmniGenerator(opts, latent_shape, verbose, no_init)
    if no_init:
        return G.to(device)
    for model in G.decoders:
        if model == "s":
            continue
        net = G.decoders[model]
        nets = [(dom, sub) for dom, sub in net.items()] if isinstance(net, nn.ModuleDict) else [("", net)]
        for dom, sub in nets:
            init_weights(sub, init_type=_opt(opts.gen[model], "init_type", "xavier"),
                         init_gain=_opt(opts.gen[model], "init_gain", 0.02), verbose=verbose,
                         caller=("create_generator decoder %s %s" % (model, dom)).strip())
    if G.encoder is not None and opts.gen.encoder.architecture == "base":
        init_weights(G.encoder, init_type=_opt(opts.gen.encoder, "init_type", "xavier"),
                     init_gain=_opt(opts.gen.encoder, "init_gain", 0.02), verbose=verbose,
                     caller="create_generator encoder")
    return G.to(device)


def _opt(node, key, default):
    """``node[key]`` with the value of ``gen.default`` (defaults.yaml:89-99, merged into every task by the reference's
    ``load_opts``, utils.py:183-188) when the key is absent or an empty auto-vivified node."""
    try:
        v = node[key]
    except (KeyError, TypeError):
        return default
    return default if (isinstance(v, dict) and not v) else v


class OmniGenerator(nn.Module):
    def __init__(self, opts, latent_shape=None, verbose=0, no_init=False):
        super().__init__()
        self.opts = opts
        self.verbose = verbose
        self.encoder = None
        self.compute_dtype = DEFAULT_COMPUTE_DTYPE
        self.pair_precision = False      # set_compute_dtype("pair16"): split-precision Masker inference
        if any(t in opts.tasks for t in "msd"):
            self.encoder = create_encoder(opts, no_init, verbose)
        decoders = {}
        if "d" in opts.tasks:
            decoders["d"] = create_depth_decoder(opts, no_init, verbose)
        if "s" in opts.tasks:
            decoders["s"] = create_segmentation_decoder(opts, no_init, verbose)
        if "m" in opts.tasks:
            decoders["m"] = create_mask_decoder(opts, no_init, verbose)
        self.decoders = nn.ModuleDict(decoders)
        self.painter = nn.Module()
        if "p" in opts.tasks:
            self.painter = create_painter(opts, no_init, verbose)

    def load_val_painter(self):
        """reference generator.py:357-411: graft a validation-only Painter from another run.  ``opts.val.val_painter``
        must be a checkpoint FILE whose run directory (two levels up) holds the ``opts.yaml`` that Painter was built
        with; its ``G`` entries are loaded with the ``painter.`` prefix stripped, the module is put in eval mode on this
        generator's device with gradients disabled.  Any failure is reported and answered with False, as in the
        reference (the trainer then simply has no painter, trainer.py:725)."""
        import traceback
        from pathlib import Path

        import yaml

        from .config import Opts
        try:
            assert self.opts.val.val_painter
            ckpt_path = Path(self.opts.val.val_painter).resolve()
            assert ckpt_path.exists()
            assert ckpt_path.is_file()
            opts_path = ckpt_path.parent.parent / "opts.yaml"
            assert opts_path.exists()
            with opts_path.open("r") as f:
                val_painter_opts = Opts(yaml.safe_load(f))
            device = next(self.parameters()).device
            state_dict = torch.load(ckpt_path, map_location=device, weights_only=False)
            painter = create_painter(val_painter_opts)
            painter.load_state_dict({k.replace("painter.", ""): v for k, v in state_dict["G"].items()})
            self.painter = painter.eval().to(device)
            self.painter.compute_dtype = self.compute_dtype
            for p in self.painter.parameters():
                p.requires_grad = False
            print("    - Loaded validation-only painter")
            return True
        except Exception as e:
            print(traceback.format_exc())
            print(e)
            print(">>> WARNING: error (^) in load_val_painter, aborting.")
            return False

    def set_compute_dtype(self, dtype):
        """torch.float16 / torch.bfloat16: the 16-bit type the kernels compute and store in.  "split24" / "pair16": the
        split-precision INFERENCE modes of the Masker -- every activation carried as several 16-bit numbers whose sum it is
        (ops.PairMap, csrc/pair.hip) through the same MFMA kernels: bf16 triples hi + mid + lo (24 bits of mantissa at any
        magnitude, 6x the multiply work: the fp32 arithmetic of the reference's default, non ``--half`` apply_events run --
        what ``G.float()`` selects, for the literal flood-mask parity north_star asks for) or fp16 pairs hi + lo (3x the
        work; 22 bits where the low part stays a normal fp16 number, an absolute floor of 2^-24 below |v| ~ 0.1).  The
        Painter and the event kernels keep running in 16 bit on the maps rounded once."""
        pair = dtype in ("pair16", "split24")
        if pair:
            why = self._pair_unsupported()
            if why:
                raise NotImplementedError("split-precision inference (%s) is not built for this generator: %s" % (dtype, why))
            if not self.pair_precision:
                self._dtype_before_pair = self.compute_dtype      # train() / a later cast restores it
            dtype = torch.float16 if dtype == "pair16" else torch.bfloat16
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float16, torch.bfloat16, \"split24\" or \"pair16\"")
        for m in self.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
            if hasattr(m, "pair_precision"):
                m.pair_precision = pair
        self.compute_dtype = dtype
        self.pair_precision = pair
        return self

    def set_painter_compute_dtype(self, dtype):
        """HYBRID inference (round 6): the Painter alone on a 16-bit type (torch.float16 / torch.bfloat16) while the Masker
        stays in the split-precision mode ``G.float()`` / ``set_compute_dtype("split24" | "pair16")`` selected -- the flood
        MASK is the fp32-grade one (it is the Masker's output alone: the bit-exact half of north_star's parity statement),
        the painted image carries the 16-bit Painter's tolerance, and the Painter's share of the work drops 6-fold (3-fold
        against "pair16").  Call it after ``set_compute_dtype``; the next ``set_compute_dtype`` / ``float()`` / ``half()``
        overrides it."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("set_painter_compute_dtype: torch.float16 or torch.bfloat16")
        for m in self.painter.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = dtype
            if hasattr(m, "pair_precision"):
                m.pair_precision = False
        return self

    def _pair_unsupported(self):
        """Why this generator cannot run the split-precision Masker (None: it can).  The ResNet encoder, the DADA depth
        decoder, the DeepLab segmentation decoder and both mask decoders (plain; SPADE since round 5) carry pair maps."""
        if self.encoder is not None and not hasattr(self.encoder, "pair_precision"):
            return "the encoder has no pair-map path"
        # (the depth / segmentation / mask decoders dispatch on the map type they are handed: nothing to check there)
        if sum(q.numel() for q in self.painter.parameters()) > 0 and not hasattr(self.painter, "pair_precision"):
            return "the painter has no pair-map path"
        return None

    def train(self, mode=True):
        """nn.Module.train; the split-precision modes are INFERENCE modes: switching to training leaves them and restores the
        16-bit type that was selected before (a training forward on pair maps is not built)."""
        if mode and self.pair_precision:
            self.set_compute_dtype(getattr(self, "_dtype_before_pair", None) or self.compute_dtype)
        return super().train(mode)

    # nn.Module's dtype casts (reference apply_events.py:467-468 ``trainer.G.half()``; trainer.py's ``.to(device)``).  The
    # parameters of this package ARE the fp32 masters -- spectral norm power-iterates them, ExtraAdam steps them and the
    # kernels read 16-bit packs made from them -- so a cast selects the 16-bit type the kernels compute and store in and
    # leaves the parameters alone; ``.float()`` on an eval-mode generator selects the split-precision Masker (below).
    def half(self):
        return self.set_compute_dtype(torch.float16)

    def bfloat16(self):
        return self.set_compute_dtype(torch.bfloat16)

    def float(self):
        """The reference's fp32 inference (apply_events without --half).  On an eval-mode generator whose Masker can run on
        pair maps this selects the split-precision mode ("split24", see set_compute_dtype; ``train()``, ``half()`` or
        ``bfloat16()`` leave it again).  Anywhere else -- training mode, an encoder without a split-precision path -- it is a
        no-op that keeps the 16-bit compute type, as before round 4 (the reference trains in fp32; this package trains in
        bf16 with fp32 masters): nothing is half-switched and nothing that ran before starts to raise."""
        if not self.training and self._pair_unsupported() is None:
            return self.set_compute_dtype("split24")
        return self

    def to(self, *args, **kwargs):
        device, dtype, non_blocking, _ = torch._C._nn._parse_to(*args, **kwargs)
        if dtype in (torch.float16, torch.bfloat16):
            self.set_compute_dtype(dtype)
        if device is not None:
            return super().to(device, non_blocking=non_blocking)
        return self

    def freeze_spectral_norm(self, frozen=True):
        """Opt-in inference mode: see ``norms.freeze_spectral_norm``."""
        from .norms import freeze_spectral_norm
        freeze_spectral_norm(self, frozen)
        return self

    def encode(self, x):
        """reference generator.py:107-118.  x: [B,3,H,W] NCHW; returns (z_high, z_low) as NHWC containers."""
        assert self.encoder is not None
        self.encoder.compute_dtype = self.compute_dtype
        return self.encoder.forward(x)

    def masker_forward(self, x, sigmoid=True):
        """Masker inference (reference trainer.py:272-287: z = encode(x); d, z_depth = dec_d(z); s = dec_s(z, z_depth);
        m = mask(z=z, z_depth=z_depth)).  Returns NCHW fp32 tensors {"d", "s", "m"} (m = sigmoid(logits) by default)."""
        z = self.encode(x)
        out = {}
        z_depth = None
        if "d" in self.decoders:
            d, z_depth = self.decoders["d"].forward_nhwc(z)
            out["d"] = Fn.to_nchw(d).to(x.dtype)
        s_nhwc = None
        if "s" in self.decoders:
            s_nhwc = self.decoders["s"].forward_nhwc(z, z_depth)
            out["s"] = Fn.to_nchw(s_nhwc).to(x.dtype)
        if "m" in self.decoders:
            cond = self.make_m_cond(d, s_nhwc, x) if self.opts.gen.m.use_spade else None     # trainer.py:285-286
            out["m"] = self.mask(z=z, cond=cond, z_depth=z_depth, sigmoid=sigmoid).to(x.dtype)
        return out

    def decode(self, x=None, z=None, return_z=False, return_z_depth=False):
        """reference generator.py:120-177 (default config: no SPADE conditioning of the mask decoder)."""
        assert x is not None or z is not None
        if z is None:
            z = self.encode(x)
        out = {}
        z_depth = None
        if "d" in self.decoders:
            d, z_depth = self.decoders["d"].forward_nhwc(z)
            out["d"] = Fn.to_nchw(d)
        if "s" in self.decoders:
            out["s"] = Fn.to_nchw(self.decoders["s"].forward_nhwc(z, z_depth))
        if "m" in self.decoders:
            out["m"] = self.mask(z=z, z_depth=z_depth)
        if return_z:
            out["z"] = z
        if return_z_depth:
            out["z_depth"] = z_depth
        return out

    def depth(self, x=None, z=None, return_z_depth=False):
        """reference generator.py:330-355"""
        assert x is not None or z is not None
        assert "d" in self.decoders
        if z is None:
            z = self.encode(x)
        d, z_depth = self.decoders["d"].forward_nhwc(z)
        d = Fn.to_nchw(d)
        return (d, z_depth) if return_z_depth else d

    def make_m_cond(self, d, s, x=None):
        """reference generator.py:196-230: cat[normalize(d), softmax(s), bilinear(x)] (x when cond_nc == 15).  d, s:
        the NHWC maps of this package's depth / segmentation decoders; returns the NHWC conditioning map."""
        # the reference hands over the decoders' NCHW predictions (trainer.py:1233-1238); this package's own callers
        # pass the NHWC maps.  A 16-bit round trip of values that came out of 16-bit maps is exact.
        dt = self.compute_dtype
        d, s = Fn.from_nchw(d, dt), Fn.from_nchw(s, dt)
        if self.opts.gen.m.spade.cond_nc == 15:
            if x is None:
                raise ValueError("When using spade for the Masker with 15 channels, x MUST be provided")
        else:
            x = None
        if (torch.is_grad_enabled() and not self.opts.gen.m.spade.detach                 # generator.py:216-218
                and (d.t.requires_grad or s.t.requires_grad)):
            from .autograd import MakeMCondFn
            cond_c = 1 + s.c + (3 if x is not None else 0)
            return ops.NHWC(MakeMCondFn.apply(d.t, s.t, x, s.c), cond_c)
        return ops.make_m_cond(ops.detached(d), ops.detached(s), x)      # spade.detach, or nothing to differentiate

    def mask(self, x=None, z=None, cond=None, z_depth=None, sigmoid=True):
        """reference generator.py:232-277: logits = decoders["m"](z, cond, z_depth); sigmoid by default."""
        assert x is not None or z is not None
        if z is None:
            z = self.encode(x)
        dec = self.decoders["m"]
        if cond is None and self.opts.gen.m.use_spade:                       # generator.py:257-262
            assert "s" in self.opts.tasks and "d" in self.opts.tasks
            d_pred, z_d = self.decoders["d"].forward_nhwc(z)
            s_pred = self.decoders["s"].forward_nhwc(z, z_d)
            cond = self.make_m_cond(d_pred, s_pred, x)
        if z_depth is None and self.opts.gen.m.use_dada:
            _, z_depth = self.decoders["d"].forward_nhwc(z)
        spectral_norm_step_all(dec, z[0].t.dtype if isinstance(z, (tuple, list)) else z.t.dtype)
        logits = dec.forward_nhwc(z, cond, z_depth)
        if sigmoid:
            logits = Fn.sigmoid(logits)
        return Fn.to_nchw(logits)

    def mask_nhwc(self, z, cond=None, z_depth=None):
        """Training-path form of ``mask``: the mask decoder's LOGITS as a differentiable NHWC map (the sigmoid / pair /
        loss kernels of ``climategan_amd.losses`` take it from there)."""
        dec = self.decoders["m"]
        spectral_norm_step_all(dec, z[0].t.dtype if isinstance(z, (tuple, list)) else z.t.dtype)
        return dec.forward_nhwc(z, cond, z_depth)

    def sample_painter_z(self, batch_size, device, force_half=False):
        """reference generator.py:179-194"""
        if self.opts.gen.p.no_z:
            return None
        z = torch.empty(batch_size, self.opts.gen.p.latent_dim, self.painter.z_h, self.painter.z_w,
                        device=device).normal_(mean=0, std=1.0)
        if force_half:
            z = z.half()
        return z

    def paint_cloudy(self, m, x, s, sky_idx=9, res=(8, 8), weight=0.8):
        """reference generator.py:299-328: the Painter is probed with an intermediary image whose sky (class ``sky_idx``
        of the bilinearly up-sampled segmentation ``s``) is replaced by Perlin clouds; the result is pasted on the
        ORIGINAL x.  ``s``: the segmentation decoder's NHWC logits (``decoders["s"].forward_nhwc``).  The lattice angles
        are drawn with ``torch.rand`` on the host like the reference does (tutils.py:660), so a seeded run sees the same
        clouds."""
        import math

        if not isinstance(s, ops.NHWC):
            raise TypeError("paint_cloudy: s must be the NHWC logits of this package's segmentation decoder")
        p = self.painter
        dt = p.compute_dtype
        angles = (2 * math.pi * torch.rand(res[0] + 1, res[1] + 1)).to(x.device)
        m = m.to(x.dtype)
        cond = ops.cloudy_cond(x, m, s, angles, sky_idx=sky_idx, weight=weight)       # noised_x * (1 - m)
        z = self.sample_painter_z(x.shape[0], x.device)
        zz = ops.nchw_to_nhwc(z, dt) if z is not None else None
        fake = p.forward_nhwc(zz, cond)                                                # paint(m, noised_x, no_paste=True)
        if fake.t.requires_grad:
            raise NotImplementedError("paint_cloudy is an inference path (call under torch.no_grad())")
        return ops.nhwc_to_nchw(fake, paste_x=x, paste_m=m).to(x.dtype)               # x * (1 - m) + fake * m

    def paint_nhwc(self, m, x):
        """Training-path form of ``paint``: returns the Painter's raw output (before the paste) as a differentiable
        NHWC map; the paste x (1 - m) + fake m and what follows it (discriminator input, VGG input) are produced by
        ``autograd.PainterHeadsFn`` so that no NCHW fp32 copy of the image is ever materialised."""
        p = self.painter
        dt = p.compute_dtype
        z = self.sample_painter_z(x.shape[0], x.device)
        cond = ops.nchw_to_nhwc(x, dt, cs=4, mask=m.to(x.dtype))        # x * (1 - m)
        zz = ops.nchw_to_nhwc(z, dt) if z is not None else None
        return p.forward_nhwc(zz, cond)

    def paint(self, m, x, no_paste=False):
        """reference generator.py:279-297: fake = painter(z, x * (1 - m)); returns x * (1 - m) + fake * m.

        m [B,1,H,W] (1 where water is painted), x [B,3,H,W] in [-1,1]; NCHW in, NCHW out (x's dtype)."""
        p = self.painter
        dt = p.compute_dtype
        z = self.sample_painter_z(x.shape[0], x.device)
        m = m.to(x.dtype)
        pair_ok = not (torch.is_grad_enabled() and any(q.requires_grad for q in p.parameters()))
        if getattr(p, "pair_precision", False) and not pair_ok and not getattr(self, "_warned_pair_grad", False):
            # (advisor, round 5: this used to switch to the 16-bit Painter without a word)
            import warnings
            warnings.warn("OmniGenerator.paint: the split-precision mode (G.float() / set_compute_dtype('split24' | 'pair16')) is an "
                          "inference mode; with autograd enabled and a trainable Painter this call runs the 16-bit Painter. Wrap "
                          "the call in torch.no_grad() for the fp32-grade result.")
            self._warned_pair_grad = True
        if getattr(p, "pair_precision", False) and pair_ok:
            # split-precision inference (round 5): the Painter on split maps -- every conv as a split-precision conv, SPADE
            # unfused with its de-normalisation in fp32 (norms.SPADE._forward_pair): the arithmetic of the reference's fp32 run
            xf, mf = x.float(), m.float()
            cond = ops.pair_from_nchw(xf * (1.0 - mf), dt)
            zz = ops.pair_from_nchw(z.float(), dt) if z is not None else None
            fake = ops.nhwc_to_nchw(p.forward_nhwc(zz, cond))
            paste = self.opts.gen.p.paste_original_content and not no_paste
            return (xf * (1.0 - mf) + fake * mf).to(x.dtype) if paste else fake.to(x.dtype)
        zz = ops.nchw_to_nhwc(z, dt) if z is not None else None
        paste = self.opts.gen.p.paste_original_content and not no_paste
        if torch.is_grad_enabled() and (m.requires_grad or x.requires_grad):
            # the mask (or the image) is itself a prediction -- painter_loss_for_masker, trainer.py:1618-1651: the
            # masking and the paste are torch expressions so that autograd carries their gradients; the Painter's
            # conditioning map then wants a gradient too (SpadeFn's cond branch, ResizeNearestFn)
            fake = Fn.to_nchw(p.forward_nhwc(zz, Fn.from_nchw(x * (1.0 - m), dt, cs=4)))
            return (x * (1.0 - m) + fake * m).to(x.dtype) if paste else fake.to(x.dtype)
        cond = ops.nchw_to_nhwc(x, dt, cs=4, mask=m)                     # x * (1 - m)
        fake = p.forward_nhwc(zz, cond)
        if paste:
            if tuple(fake.t.shape[1:3]) != tuple(x.shape[-2:]):
                raise RuntimeError("paint: painter output %s does not match x %s (input must be a multiple of %d)"
                                   % (tuple(fake.t.shape[1:3]), tuple(x.shape[-2:]), 2 ** p.spade_n_up))
            return Fn.to_nchw(fake, paste_x=x, paste_m=m).to(x.dtype)
        return Fn.to_nchw(fake).to(x.dtype)
