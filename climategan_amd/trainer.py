"""Host-side mirror of the inference half of the reference's ``climategan/trainer.py``:
``Trainer.setup(inference=True)``, ``infer_all`` and ``compute_flood`` -- the stage order, binarisation and uint8
conversion of the reference, every arithmetic step a HIP kernel behind the C ABI.

Built: the flood event (Masker -> mask -> Painter, optionally through ``paint_cloudy``), the smog event (depth -> HazeRD
transmission model) and the wildfire event (``fire.add_fire``; its torchvision / kornia arithmetic is restated from those
libraries' documentation because they are not in the reference tree, so that event is pinned by the oracle only).

Training half (row H2), Painter tasks only (``opts.tasks == ["p"]``): ``setup(inference=False)`` builds G, D, the
losses and the two ExtraAdam optimisers; ``update_G`` / ``update_D`` / ``train_step`` reproduce the "rf" branch of
``get_painter_loss`` (trainer.py:1256-1387) and ``get_D_loss`` (trainer.py:1073-1107) and the extrapolate / step
schedule (trainer.py:674-694).  Masker domains raise (training-mode BatchNorm and the masker losses are not built).
"""
import time

import torch

from . import ops
from .generator import create_generator
from .utils import find_target_size


class Timer:
    """reference utils.py:899-960: context manager appending elapsed seconds to ``store`` (device-synchronised)."""

    def __init__(self, name="", store=None, precision=3, ignore=False, cuda=True):
        self.store = store
        self.cuda = cuda and torch.cuda.is_available()
        self.ignore = ignore

    def __enter__(self):
        if not self.ignore:
            if self.cuda:
                torch.cuda.synchronize()
            self._t = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if not self.ignore:
            if self.cuda:
                torch.cuda.synchronize()
            if self.store is not None:
                self.store.append(time.perf_counter() - self._t)
        return False


class Trainer:
    """Inference-side subset of the reference Trainer (trainer.py:63-216): owns ``G``; no logger / comet / data."""

    def __init__(self, opts, comet_exp=None, verbose=0, device=None):
        self.opts = opts
        self.verbose = verbose
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda:0" if torch.cuda.is_available() else "cpu")
        self.G = None
        self.D = None
        self.is_setup = False
        self.has_painter = "p" in opts.tasks

    def setup(self, inference=False):
        """reference trainer.py:701-789."""
        self.G = create_generator(self.opts, device=self.device, no_init=inference, verbose=self.verbose)
        if self.has_painter:
            self.G.painter.set_latent_shape(find_target_size(self.opts, "x"), True)       # trainer.py:727-728
        if inference:
            self.G.eval()
            self.is_setup = True
            return self
        if list(self.opts.tasks) != ["p"]:
            raise NotImplementedError("Trainer.setup(inference=False): only the Painter tasks (opts.tasks == ['p']) "
                                      "have a HIP training path; Masker training (training-mode BatchNorm, masker "
                                      "losses) is not built")
        from .discriminator import create_discriminator
        from .losses import get_losses
        from .optim import ExtraAdam

        o = self.opts
        self.D = create_discriminator(o, self.device, verbose=self.verbose)
        self.G.train()
        self.D.train()
        # get_losses (losses.py:353-441).  Note: losses["D"]["p"] IS losses["G"]["p"]["gan"] (one GANLoss object), so
        # the generator-side call draws label smoothing / flips as well, as in the reference.
        self.losses = get_losses(o, self.verbose, self.device)
        if o.train.lambdas.G.p.vgg == 0:
            self.losses["G"]["p"]["vgg"] = None
        g_params = [p for p in self.G.parameters() if p.requires_grad]
        d_params = [p for p in self.D.parameters() if p.requires_grad]
        self.g_opt = ExtraAdam(g_params, lr=o.gen.opt.lr.default, betas=(o.gen.opt.beta1, 0.999))
        self.d_opt = ExtraAdam(d_params, lr=o.dis.opt.lr.default, betas=(o.dis.opt.beta1, 0.999))
        self.global_step = 0
        self.loss_log = {}
        # data parallel (one process per GPU, launched by torchrun): identical replicas, bucketed gradient all-reduce
        # overlapped with the backward
        from .parallel import GradBucketReducer, broadcast_parameters, is_distributed
        self.g_reducer = self.d_reducer = None
        if is_distributed():
            broadcast_parameters(self.G)
            broadcast_parameters(self.D)
            self.g_reducer = GradBucketReducer(g_params)
            self.d_reducer = GradBucketReducer(d_params)
        self.is_setup = True
        return self

    # ------------------------------------------------------------------------------------------ training
    def _painter_terms(self, batch, for_g):
        """D(cat_batch[real, fake]) on NHWC inputs built by the heads kernel; returns (real_d, fake_d, vgg pair)."""
        from .autograd import PainterHeadsFn
        from .tutils import divide_pred

        x, m = batch["data"]["x"], batch["data"]["m"]
        dt = self.G.painter.compute_dtype
        want_vgg = for_g and self.losses["G"]["p"]["vgg"] is not None
        if for_g:
            fake = self.G.paint_nhwc(m, x)
        else:
            with torch.no_grad():                                     # trainer.py:1076-1083
                fake = self.G.paint_nhwc(m, x)
        real_in, vgg_real = ops.painter_heads(None, x, m, dt, True, want_vgg)
        if for_g:
            fake_in_t, vgg_fake_t = PainterHeadsFn.apply(fake.t, x, m, True, want_vgg)
        else:
            fake_in, _ = ops.painter_heads(fake, x, m, dt, True, False)
            fake_in_t, vgg_fake_t = fake_in.t, None
        real_fake_cat = ops.NHWC(torch.cat([real_in.t, fake_in_t], dim=0), 4)            # trainer.py:1103,1363
        real_fake_d = self.D["p"](real_fake_cat, nhwc=True)
        real_d, fake_d = divide_pred(real_fake_d)
        vgg = (ops.NHWC(vgg_fake_t, 3), vgg_real) if want_vgg else None
        return real_d, fake_d, vgg

    def get_painter_loss(self, multi_domain_batch):
        """reference trainer.py:1256-1387 (single-discriminator branch; TV / context / reconstruction lambdas are 0 in
        defaults.yaml:293-300 and have no HIP kernel: non-zero values raise)."""
        lambdas = self.opts.train.lambdas
        for k in ("tv", "context", "reconstruction"):
            if lambdas.G.p[k] != 0:
                raise NotImplementedError("painter loss '%s' has no HIP kernel (lambda must be 0)" % k)
        real_d, fake_d, vgg = self._painter_terms(multi_domain_batch["rf"], True)
        step_loss = 0
        if vgg is not None:
            loss = self.losses["G"]["p"]["vgg"](vgg[0], vgg[1]) * lambdas.G.p.vgg
            self.loss_log["G.p.vgg"] = loss.detach()
            step_loss = step_loss + loss
        loss = self.losses["G"]["p"]["gan"](fake_d, True, False)        # not scaled by lambdas.G.p.gan (trainer.py:1369-1371)
        self.loss_log["G.p.gan"] = loss.detach()
        step_loss = step_loss + loss
        if self.opts.dis.p.get_intermediate_features and lambdas.G.p.featmatch != 0:
            loss = self.losses["G"]["p"]["featmatch"](real_d, fake_d) * lambdas.G.p.featmatch
            self.loss_log["G.p.featmatch"] = loss.detach()
            step_loss = step_loss + loss
        return step_loss

    def get_D_loss(self, multi_domain_batch):
        """reference trainer.py:1034-1160, Painter branch (1073-1107)."""
        real_d, fake_d, _ = self._painter_terms(multi_domain_batch["rf"], False)
        loss = self.losses["D"]["p"](fake_d, False, True)
        loss = loss + self.losses["D"]["p"](real_d, True, True)
        self.loss_log["D.p.gan"] = loss.detach()
        return loss

    def _check_batch(self, multi_domain_batch):
        extra = [d for d in multi_domain_batch if d != "rf"]
        if extra:
            raise NotImplementedError("Trainer: masker domains %s have no HIP training path yet" % extra)

    def update_G(self, multi_domain_batch):
        """reference trainer.py:989-1015 + g_opt_step (674-683): D frozen, backward, extrapolate (even) / step (odd)."""
        self._check_batch(multi_domain_batch)
        for p in self.D.parameters():                                   # trainer.py:959-962
            p.requires_grad_(False)
        try:
            self.g_opt.zero_grad(set_to_none=True)
            g_loss = self.get_painter_loss(multi_domain_batch)
            g_loss.backward()
            if self.g_reducer is not None:
                self.g_reducer.finish()                                 # before extrapolation AND step (trainer.py:678-683)
            if self.global_step % 2 == 0:
                self.g_opt.extrapolation()
            else:
                self.g_opt.step()
        finally:
            self._restore_d_grad_flags()                                # trainer.py:971-973
        return g_loss.detach()

    def _restore_d_grad_flags(self):
        for name, p in self.D.named_parameters():
            p.requires_grad_(not (name.endswith("weight_u") or name.endswith("weight_v")))

    def update_D(self, multi_domain_batch):
        """reference trainer.py:1017-1032 + d_opt_step (685-694)."""
        self._check_batch(multi_domain_batch)
        self.d_opt.zero_grad(set_to_none=True)
        d_loss = self.get_D_loss(multi_domain_batch)
        d_loss.backward()
        if self.d_reducer is not None:
            self.d_reducer.finish()
        if self.global_step % 2 == 0:
            self.d_opt.extrapolation()
        else:
            self.d_opt.step()
        return d_loss.detach()

    def train_step(self, multi_domain_batch):
        """One iteration of run_epoch's loop body (trainer.py:939-976): G update, D update, step counter."""
        g = self.update_G(multi_domain_batch)
        d = self.update_D(multi_domain_batch)
        self.global_step += 1
        return g, d

    # ------------------------------------------------------------------------------------------ events
    def compute_flood(self, x, z=None, z_depth=None, m=None, s=None, cloudy=None, bin_value=-1):
        """reference trainer.py:1844-1877"""
        if m is None:
            if z is None:
                z = self.G.encode(x)
            if "d" in self.opts.tasks and self.opts.gen.m.use_dada and z_depth is None:
                _, z_depth = self.G.decoders["d"].forward_nhwc(z)
            m = self.G.mask(x=x, z=z, z_depth=z_depth)
        if bin_value >= 0:
            m = ops.binarize(m, bin_value)                                                # (m > bin_value).to(m.dtype)
        if cloudy:
            assert s is not None
            return self.G.paint_cloudy(m, x, s)
        return self.G.paint(m, x)

    def compute_fire(self, x, seg_preds=None, z=None, z_depth=None):
        """reference trainer.py:1821-1842 -> fire.add_fire (fire.py:68-126).  ``seg_preds``: the segmentation decoder's
        NHWC logits or None.  The filter's green level is ``random.randint(100, 150)`` like the reference's (fire.py:115)."""
        import random

        if seg_preds is None:
            if z is None:
                z = self.G.encode(x)
            seg_preds = self.G.decoders["s"].forward_nhwc(z, z_depth)
        if not isinstance(seg_preds, ops.NHWC):
            raise TypeError("compute_fire: seg_preds must be the NHWC logits of this package's segmentation decoder")
        f = self.opts.events.fire
        out = ops.wildfire(x, seg_preds, float(random.randint(100, 150)), kernel_size=f.get("kernel_size", 301),
                           kernel_sigma=f.get("kernel_sigma", 150.5), transparency=200,
                           crop_bottom=bool(f.get("crop_bottom_sky_mask")))
        return out.to(x.dtype)

    def compute_smog(self, x, z=None, d=None, s=None, use_sky_seg=False):
        """reference trainer.py:1879-1939 (``use_sky_seg`` is a no-op there too: the sky mask is never built).
        ``d``: the depth decoder's NHWC map (``G.decoders["d"].forward_nhwc``) or None."""
        if d is None:
            if z is None:
                z = self.G.encode(x)
            d, _ = self.G.decoders["d"].forward_nhwc(z)
        if not isinstance(d, ops.NHWC):
            raise TypeError("compute_smog: d must be the NHWC depth map of this package's depth decoder")
        prm = self.opts.events.smog
        out = ops.smog(x, d, prm.airlight, prm.beta / prm.vr, prm.alpha / 255.0, [v / 255.0 for v in prm.yellow_color])
        return out.to(x.dtype)

    @torch.no_grad()
    def infer_all(self, x, numpy=True, stores={}, bin_value=-1, half=False, xla=False, cloudy=False,
                  auto_resize_640=False, ignore_event=set(), return_masks=False):
        """reference trainer.py:217-334.  ``half`` selects fp16 I/O tensors (the kernels compute in 16-bit either
        way); ``xla`` is accepted and ignored.  Events not in ``ignore_event`` must have a HIP path."""
        assert self.is_setup
        assert len(x.shape) in {3, 4}, f"Unknown Data shape {x.shape}"
        if not isinstance(x, torch.Tensor):
            x = torch.tensor(x, device=self.device)
        if len(x.shape) == 3:
            x = x.unsqueeze(0)
        if x.shape[1] != 3:
            assert x.shape[-1] == 3, f"Unknown x shape to permute {x.shape}"
            x = x.permute(0, 3, 1, 2)
        if x.device != self.device:
            x = x.to(self.device)
        if auto_resize_640 and (x.shape[-1] != 640 or x.shape[-2] != 640):
            raise NotImplementedError("auto_resize_640: the input-side resize (SURVEY row N3) has no HIP path yet")
        x = x.half() if half else x.float()
        x = x.contiguous()

        self.G.painter.set_latent_shape(x.shape, True)                                   # trainer.py:266

        with Timer(store=stores.get("all events", [])):
            with Timer(store=stores.get("encode", [])):
                z = self.G.encode(x)
            with Timer(store=stores.get("depth", [])):
                depth_nhwc, z_depth = self.G.decoders["d"].forward_nhwc(z)
            with Timer(store=stores.get("segmentation", [])):
                seg_nhwc = self.G.decoders["s"].forward_nhwc(z, z_depth)
            with Timer(store=stores.get("mask", [])):
                cond = self.G.make_m_cond(depth_nhwc, seg_nhwc, x) if self.opts.gen.m.use_spade else None   # :285
                mask = self.G.mask(z=z, cond=cond, z_depth=z_depth).to(x.dtype)

            wildfire = smog = flood = None
            if "wildfire" not in ignore_event:
                with Timer(store=stores.get("wildfire", [])):
                    wildfire = self.compute_fire(x, seg_preds=seg_nhwc)
            if "smog" not in ignore_event:
                with Timer(store=stores.get("smog", [])):
                    smog = self.compute_smog(x, d=depth_nhwc, s=seg_nhwc)
            if "flood" not in ignore_event:
                with Timer(store=stores.get("flood", [])):
                    flood = self.compute_flood(x, m=mask, s=seg_nhwc, cloudy=cloudy, bin_value=bin_value)

        output_data = {}
        with Timer(store=stores.get("numpy", []), ignore=not numpy):
            for name, ev in (("flood", flood), ("wildfire", wildfire), ("smog", smog)):
                if ev is None:
                    continue
                if numpy:
                    ev = ops.normalize_to_uint8(ev).cpu().numpy()                        # trainer.py:311-326
                output_data[name] = ev
        if return_masks:
            output_data["mask"] = ops.binarize(mask, bin_value, want_float=False, want_uint8=True).cpu().numpy()
        return output_data
