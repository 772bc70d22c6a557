"""Host-side mirror of the inference half of the reference's ``climategan/trainer.py``:
``Trainer.setup(inference=True)``, ``infer_all`` and ``compute_flood`` -- the stage order, binarisation and uint8
conversion of the reference, every arithmetic step a HIP kernel behind the C ABI.

Built: the flood event (Masker -> mask -> Painter, optionally through ``paint_cloudy``), the smog event (depth -> HazeRD
transmission model) and the wildfire event (``fire.add_fire``; its torchvision / kornia arithmetic is restated from those
libraries' documentation because they are not in the reference tree, so that event is pinned by the oracle only).

Training half (row H2): ``setup(inference=False)`` builds G, D, the losses, the two ExtraAdam optimisers and their
schedulers (``get_optimizer``); ``update_G`` / ``update_D`` / ``train_step`` reproduce ``get_masker_loss``
(trainer.py:1184-1254, 1389-1616), ``get_painter_loss`` (trainer.py:1256-1387), ``get_D_loss`` (trainer.py:1034-1160)
and the extrapolate / step schedule (trainer.py:674-694); ``save`` / ``resume`` / ``update_learning_rates`` are the
checkpoint half (trainer.py:396-579): same file layout, same path rules.
"""
import os
import time

import torch

from . import autograd as ag
from . import ops
from .generator import create_generator
from .utils import find_target_size

_Z_PASS = os.environ.get("CGAN_Z_PASS", "1") != "0"     # same-box A/B: the latent's fan-in summed by the autograd engine


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


def _merge(source, destination):
    """reference utils.py:68-105: recursive dict merge, ``source`` entries overwrite ``destination``'s."""
    for key, value in source.items():
        if isinstance(value, dict):
            node = destination.setdefault(key, {})
            _merge(value, node)
        else:
            destination[key] = value
    return destination


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
        self.use_pl4m = False            # trainer.py:94; maybe_enable_pl4m() / run_epoch() switch it on at opts.gen.p.pl4m_epoch
        # the real and the simulated domain batch share the Masker's encoder / depth / segmentation launches (grouped
        # BatchNorm keeps the per-domain statistics of the reference's separate calls); False = one pass per domain
        self.merge_domains = True
        # the Masker branch and the Painter branch of an update (disjoint parameters, independent until the optimizer
        # step) are issued on two HIP streams: the tail of one kernel overlaps the head of the other branch's next one
        # instead of every launch draining the chip on its own (CGAN_OVERLAP=0: one stream, same-box A/B)
        import os
        self.overlap_branches = os.environ.get("CGAN_OVERLAP", "1") != "0"
        self._side = None
        # development aid (tools/branch_times.py): [(fork event, end of the main-stream branch, end of the side-stream branch)]
        self.branch_events = [] if os.environ.get("CGAN_BRANCH_TIMES") == "1" else None

    def setup(self, inference=False):
        """reference trainer.py:701-789."""
        self.G = create_generator(self.opts, device=self.device, no_init=inference, verbose=self.verbose)
        # trainer.py:725: a generator without its own Painter may borrow a validation-only one (opts.val.val_painter)
        own_painter = sum(p.numel() for p in self.G.painter.parameters()) > 0
        self.has_painter = own_painter or (str(self.opts.val.val_painter) != "none" and self.G.load_val_painter())
        if self.has_painter:
            self.G.painter.set_latent_shape(find_target_size(self.opts, "x"), True)       # trainer.py:727-728
        if inference:
            if self.opts.train.resume:                                                    # trainer.py:735-736
                self.resume(True)
            self.G.eval()
            self.is_setup = True
            return self
        from .discriminator import create_discriminator
        from .losses import get_losses
        from .optim import get_optimizer

        o = self.opts
        self.D = create_discriminator(o, self.device, verbose=self.verbose)
        self.G.train()
        self.D.train()
        # 16-bit activation GRADIENTS: the masker's loss weights (1e-3 / (n h w)) underflow fp16; bf16 has the range
        self.G.set_compute_dtype(torch.bfloat16)
        self.D.set_compute_dtype(torch.bfloat16)
        self.domain_labels = {"s": 0, "r": 1}                          # trainer.py:107
        self.has_masker = any(t in o.tasks for t in "msd")
        # get_losses (losses.py:353-441).  Note: losses["D"]["p"] IS losses["G"]["p"]["gan"] (one GANLoss object), so
        # the generator-side call draws label smoothing / flips as well, as in the reference.
        self.losses = get_losses(o, self.verbose, self.device)
        if o.train.lambdas.G.p.vgg == 0:
            self.losses["G"]["p"]["vgg"] = None
        g_params = [p for p in self.G.parameters() if p.requires_grad]
        d_params = [p for p in self.D.parameters() if p.requires_grad]
        # get_optimizer (trainer.py:758-767): groups / parameter order of the reference, so g_opt / d_opt state dicts are
        # interchangeable with its checkpoints
        self.lr_names = {}
        self.g_opt, self.g_scheduler, self.lr_names["G"] = get_optimizer(self.G, o.gen.opt, o.tasks)
        self.d_opt, self.d_scheduler, self.lr_names["D"] = get_optimizer(self.D, o.dis.opt, o.tasks, True)
        self.global_step = 0                                           # reference: self.logger.global_step
        self.epoch = 0                                                 # reference: self.logger.epoch
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
        if os.environ.get("CGAN_GC_FREEZE", "0") == "1":
            self.freeze_host_objects()
        self.is_setup = True
        return self

    def freeze_host_objects(self):
        """Opt-in, for a process that trains ONE trainer for its whole life (bench.py, a training entry point; or
        CGAN_GC_FREEZE=1): everything built so far -- modules, parameters, packed-weight caches, optimizer state, ~10^6
        Python objects -- leaves the garbage collector's generations.  Left there, every full collection walks all of it:
        110-125 ms of host time every ~34 train steps (tools/step_times.py), five times the lead the host has over the
        GPU.  The side effect is process-global (``gc.freeze()`` moves EVERY live object, the caller's too, to the
        permanent generation, where reference cycles are never reclaimed), which is why ``setup()`` does not do it on its
        own (advisor, round 4); ``close()`` undoes it."""
        import gc
        gc.collect()
        gc.freeze()
        self._gc_frozen = True
        return self

    def close(self):
        """Undo ``freeze_host_objects`` (the permanent generation returns to the collector)."""
        if getattr(self, "_gc_frozen", False):
            import gc
            gc.unfreeze()
            self._gc_frozen = False

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
            self._last_fake = fake                                     # (the optional image-space terms read it)
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
        vgg = (ops.NHWC(vgg_fake_t, 6), vgg_real) if want_vgg else None
        return real_d, fake_d, vgg

    def _painter_aux_terms(self, raw, x, m):
        """TV(p m) / context / reconstruction terms of reference trainer.py:1289-1315 on the painted image p, from ONE fused
        kernel (autograd.PainterAuxFn) that forms p = x (1 - m) + raw m itself: ``raw`` is the Painter's map before the paste.
        The reference applies the terms to ``G.paint()``'s output, which is the un-pasted map when
        ``gen.p.paste_original_content`` is off: that combination has no kernel and raises instead of computing another loss."""
        from .autograd import PainterAuxFn
        lam = self.opts.train.lambdas.G.p
        if not self.opts.gen.p.paste_original_content or raw is None:
            raise NotImplementedError("painter TV / context / reconstruction terms with gen.p.paste_original_content = False "
                                      "have no HIP path (the fused kernel evaluates them on the pasted image)")
        loss, parts = PainterAuxFn.apply(raw.t, x, m, float(lam.tv), float(lam.context), float(lam.reconstruction))
        for i, k in enumerate(("tv", "context", "reconstruction")):
            if lam[k] != 0:
                self.loss_log["G.p." + k] = parts[i]
        return loss

    def _painter_loss_local_pair(self, batch):
        """``dis.p.use_local_discriminator`` (off in defaults.yaml:227), G side: reference trainer.py:1323-1358.  D["p"] is
        the pair {"global", "local"} of 3-channel discriminators (discriminator.py:246-252); the global one sees the
        painted image, the local one ``fake * m``; both GAN terms are scaled by lambdas.G.p.gan (the single-discriminator
        branch is not, :1369), feature matching on the global one only, and WITHOUT the featmatch != 0 test of the other
        branch.  Written on the NCHW boundary (G.paint / D(x) / the loss classes carry their graphs): a non-default
        branch, not a fused path."""
        from .tutils import vgg_preprocess

        lambdas = self.opts.train.lambdas
        x, m = batch["data"]["x"], batch["data"]["m"]
        aux = any(lambdas.G.p[k] != 0 for k in ("tv", "context", "reconstruction"))
        if aux and self.opts.gen.p.paste_original_content:
            # the Painter's raw map is needed as well (the fused image-space kernel pastes it itself): G.paint's two halves
            from . import functional as Fn
            raw = self.G.paint_nhwc(m, x)
            fake = Fn.to_nchw(raw, paste_x=x, paste_m=m.to(x.dtype)).to(x.dtype)
        else:
            raw, fake = None, self.G.paint(m, x)
        step_loss = 0
        if lambdas.G.p.vgg != 0 and self.losses["G"]["p"]["vgg"] is not None:                       # :1276-1287
            loss = self.losses["G"]["p"]["vgg"](vgg_preprocess(fake * m), vgg_preprocess(x * m)) * lambdas.G.p.vgg
            self.loss_log["G.p.vgg"] = loss.detach()
            step_loss = step_loss + loss
        if aux:                                                                                       # :1289-1315
            step_loss = step_loss + self._painter_aux_terms(raw, x, m)
        fake_d_global = self.D["p"]["global"](fake)
        fake_d_local = self.D["p"]["local"](fake * m)
        real_d_global = self.D["p"]["global"](x)
        loss = self.losses["G"]["p"]["gan"](fake_d_global, True, False)
        loss = loss + self.losses["G"]["p"]["gan"](fake_d_local, True, False)
        loss = loss * lambdas.G["p"]["gan"]
        self.loss_log["G.p.gan"] = loss.detach()
        step_loss = step_loss + loss
        if self.opts.dis.p.get_intermediate_features:
            loss = self.losses["G"]["p"]["featmatch"](real_d_global, fake_d_global) * lambdas.G["p"]["featmatch"]
            self.loss_log["G.p.featmatch"] = loss.detach() if torch.is_tensor(loss) else loss
            step_loss = step_loss + loss
        return step_loss

    def get_painter_loss(self, multi_domain_batch):
        """reference trainer.py:1256-1387 (the TV / context / reconstruction terms, lambdas 0 in defaults.yaml:293-300, come
        from one fused kernel on the pasted image, autograd.PainterAuxFn)."""
        lambdas = self.opts.train.lambdas
        if self.opts.gen.p.get("diff_aug", {}).get("use", False):
            raise NotImplementedError("gen.p.diff_aug (transforms.py:609 DiffTransforms: data augmentation, out of scope) "
                                      "has no HIP path")
        if self.opts.dis.p.use_local_discriminator:
            return self._painter_loss_local_pair(multi_domain_batch["rf"])
        real_d, fake_d, vgg = self._painter_terms(multi_domain_batch["rf"], True)
        step_loss = 0
        if any(lambdas.G.p[k] != 0 for k in ("tv", "context", "reconstruction")):      # trainer.py:1289-1315 (0 by default)
            data = multi_domain_batch["rf"]["data"]
            step_loss = step_loss + self._painter_aux_terms(self._last_fake, data["x"], data["m"])
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
        if self.opts.gen.p.get("diff_aug", {}).get("use", False):
            raise NotImplementedError("gen.p.diff_aug has no HIP path")
        if self.opts.dis.p.use_local_discriminator:                                   # :1085-1099
            data = multi_domain_batch["rf"]["data"]
            x, m = data["x"], data["m"]
            with torch.no_grad():
                fake = self.G.paint(m, x)
            fake = fake.detach()
            crit = self.losses["D"]["p"]
            g_loss = crit(self.D["p"]["global"](fake), False, True) + crit(self.D["p"]["global"](x), True, True)
            l_loss = crit(self.D["p"]["local"](fake * m), False, True) + crit(self.D["p"]["local"](x * m), True, True)
            self.loss_log["D.p.global"], self.loss_log["D.p.local"] = g_loss.detach(), l_loss.detach()
            return g_loss + l_loss
        real_d, fake_d, _ = self._painter_terms(multi_domain_batch["rf"], False)
        loss = self.losses["D"]["p"](fake_d, False, True)
        loss = loss + self.losses["D"]["p"](real_d, True, True)
        self.loss_log["D.p.gan"] = loss.detach()
        return loss

    # ------------------------------------------------------------------------------------------ masker losses
    def masker_d_loss(self, x, z, target, domain, for_="G", pre=None):
        """reference trainer.py:1389-1407.  The reference evaluates the depth loss and then discards it for real-domain
        batches; here it is only evaluated where it is kept.  ``pre``: (prediction, z_depth) already computed by the
        merged-domain trunk."""
        prediction, z_depth = pre if pre is not None else self.G.decoders["d"].forward_nhwc(z)
        weight = self.opts.train.lambdas.G.d.main
        if weight == 0 or domain == "r":
            return torch.zeros((), device=self.device), prediction, z_depth
        loss = self.losses["G"]["tasks"]["d"](prediction, target) * weight
        self.loss_log["G.d." + domain] = loss.detach()
        return loss, prediction, z_depth

    def masker_s_loss(self, x, z, depth_preds, z_depth, target, domain, for_="G", pre=None):
        """reference trainer.py:1409-1504 (``pre``: the segmentation logits already computed by the merged-domain trunk)"""
        from . import losses as L

        assert for_ in {"G", "D"} and domain in {"r", "s"}
        o = self.opts
        full_loss = 0
        softmax_preds = None
        pred = None
        if pre is not None:
            pred = pre
        elif for_ == "G" or o.gen.s.use_advent:
            pred = self.G.decoders["s"].forward_nhwc(z, z_depth)
        if for_ == "G":
            if domain == "s":
                w = o.train.lambdas.G["s"]["crossent"]
                if w != 0:
                    loss = self.losses["G"]["tasks"]["s"]["crossent"](pred, target.squeeze(1)) * w
                    self.loss_log["G.s.crossent." + domain] = loss.detach()
                    full_loss = full_loss + loss
            if domain == "r":
                w = o.train.lambdas.G["s"]["minent"]
                if o.gen.s.get("use_minent", True) and w != 0:
                    softmax_preds = L.softmax(pred)
                    loss = self.losses["G"]["tasks"]["s"]["minent"](softmax_preds) * w
                    self.loss_log["G.s.minent.r"] = loss.detach()
                    full_loss = full_loss + loss
        if o.gen.s.use_advent:
            dp = None
            if o.gen.s.use_dada and depth_preds is not None:
                dp = ops.NHWC(depth_preds.t.detach(), depth_preds.c)
            if for_ == "D":
                label, loss_func, w = domain, self.losses["D"]["advent"], o.train.lambdas.advent.adv_main
                pred = ops.NHWC(pred.t.detach(), pred.c)
                softmax_preds = None
            else:
                label, loss_func, w = "s", self.losses["G"]["tasks"]["s"]["advent"], o.train.lambdas.G["s"]["advent"]
            if (for_ == "D" or domain == "r") and w != 0:
                # the discriminator's input is computed from the logits (losses.advent_input): same function as
                # prob_2_entropy(softmax(pred)) * depth, trainer.py:1433, 1455-1456
                loss = loss_func(softmax_preds, self.domain_labels[label], self.D["s"]["Advent"], dp, logits=pred) * w
                self.loss_log["%s.s.advent.%s" % (for_, domain)] = loss.detach()
                full_loss = full_loss + loss
        return full_loss, pred

    def masker_m_loss(self, x, z, target, domain, for_="G", cond=None, z_depth=None, depth_preds=None):
        """reference trainer.py:1506-1616"""
        from . import losses as L

        assert for_ in {"G", "D"} and domain in {"r", "s"}
        o = self.opts
        full_loss = 0
        logits = self.G.mask_nhwc(z, cond=cond, z_depth=z_depth)
        if for_ == "D":
            logits = ops.NHWC(logits.t.detach(), logits.c)
        prob = L.sigmoid_pair(logits)                                   # cat[p, 1 - p]
        if for_ == "G":
            pred_prob = L.sigmoid(logits)
            w = o.train.lambdas.G.m.tv
            if w != 0:
                loss = self.losses["G"]["tasks"]["m"]["tv"](pred_prob) * w
                self.loss_log["G.m.tv." + domain] = loss.detach()
                full_loss = full_loss + loss
            w = o.train.lambdas.G.m.bce
            if domain == "s" and w != 0:
                loss = self.losses["G"]["tasks"]["m"]["bce"](logits, target) * w
                self.loss_log["G.m.bce.s"] = loss.detach()
                full_loss = full_loss + loss
            if domain == "r":
                w = o.train.lambdas.G["m"]["gi"]
                if o.gen.m.use_ground_intersection and w != 0:
                    loss = self.losses["G"]["tasks"]["m"]["gi"](pred_prob, target) * w
                    self.loss_log["G.m.gi.r"] = loss.detach()
                    full_loss = full_loss + loss
                w = o.train.lambdas.G.m.pl4m
                if self.use_pl4m and w != 0:                                              # trainer.py:1548-1554
                    loss = self.painter_loss_for_masker(x, pred_prob) * w
                    self.loss_log["G.m.pl4m.r"] = loss.detach()
                    full_loss = full_loss + loss
                w = o.train.lambdas.advent.ent_main
                if o.gen.m.use_minent and w != 0:
                    loss = self.losses["G"]["tasks"]["m"]["minent"](prob) * w
                    self.loss_log["G.m.minent.r"] = loss.detach()
                    full_loss = full_loss + loss
        if o.gen.m.use_advent:
            dp = None
            if o.gen.m.use_dada and depth_preds is not None:
                # trainer.py:1566-1570: the detached depth prediction, nearest-resized to the image, weights the entropy map
                dp = ops.resize_nearest(ops.detached(depth_preds), (x.shape[-2], x.shape[-1]))
            if for_ == "D":
                label, loss_func = domain, self.losses["D"]["advent"]
            else:
                label, loss_func = "s", self.losses["G"]["tasks"]["m"]["advent"]
            w = o.train.lambdas.advent.adv_main
            if (for_ == "D" or domain == "r") and w != 0:
                loss = loss_func(prob, self.domain_labels[label], self.D["m"]["Advent"], dp, logits=logits,
                                 sigmoid_pair=True) * w
                self.loss_log["%s.m.advent.%s" % (for_, domain)] = loss.detach()
                full_loss = full_loss + loss
        return full_loss, prob

    def painter_loss_for_masker(self, x, m):
        """reference trainer.py:1618-1651 (pl4m; both discriminator branches): the GAN term of the Painter's discriminator
        on ``paint(m, x)`` with the PREDICTED mask probability ``m`` -- the Painter (and D, frozen throughout the G update)
        is not updated; the gradient reaches the Masker through the paste, through the discriminator's mask channel and
        through the Painter's conditioning image x (1 - m) (``SpadeFn``'s cond branch).  ``m``: NHWC map or NCHW tensor."""
        from . import functional as Fn
        from .tutils import divide_pred

        frozen = [p for p in self.G.painter.parameters() if p.requires_grad]
        for p in frozen:
            p.requires_grad_(False)
        try:
            m = Fn.to_nchw(m) if isinstance(m, ops.NHWC) else m
            fake = self.G.paint(m, x)
            if self.opts.dis.p.use_local_discriminator:                                   # trainer.py:1628-1636
                gan = self.losses["G"]["p"]["gan"]
                return gan(self.D["p"]["global"](fake), True, False) + gan(self.D["p"]["local"](fake * m), True, False)
            real_fake_cat = torch.cat([torch.cat([m, x], dim=1), torch.cat([m, fake], dim=1)], dim=0)
            _, fake_d = divide_pred(self.D["p"](real_fake_cat, nhwc=True))
            return self.losses["G"]["p"]["gan"](fake_d, True, False)
        finally:
            if "p" in self.opts.tasks:
                for p in frozen:
                    p.requires_grad_(True)

    def _merged_masker_domains(self, multi_domain_batch):
        """The masker domains of a batch that can go through the encoder and the depth / segmentation decoders as ONE
        concatenated batch (``autograd.bn_groups``: every BatchNorm keeps per-domain batch statistics and updates its
        running statistics domain after domain, so the arithmetic is that of the reference's per-domain calls,
        trainer.py:1200-1254): same image shapes, at least two domains, and ``self.merge_domains``.  None otherwise."""
        doms = [d for d in multi_domain_batch if d != "rf"]
        if not getattr(self, "merge_domains", True) or len(doms) < 2:
            return None
        shapes = {tuple(multi_domain_batch[d]["data"]["x"].shape) for d in doms}
        return doms if len(shapes) == 1 else None

    def _masker_trunk(self, multi_domain_batch, doms, want_s):
        """encode + depth decoder (+ segmentation decoder) on the concatenated domain batches; returns per-domain lists
        (z, d_pred, z_depth, s_pred) of NHWC maps."""
        from .autograd import bn_groups, split_batch

        G = len(doms)
        x = torch.cat([multi_domain_batch[d]["data"]["x"] for d in doms], dim=0)
        with bn_groups(G):
            z = self.G.encode(x)
            d_pred = z_depth = s_pred = None
            if "d" in self.opts.tasks:
                if _Z_PASS and torch.is_grad_enabled() and z[0].t.requires_grad and self.G.decoders["d"].enc4_1.norm.training:
                    # the latent's other readers take it from the depth decoder's first conv node (depth.py forward_nhwc)
                    d_pred, z_depth, z_pass = self.G.decoders["d"].forward_nhwc(z, passthrough=True)
                    z = (z_pass,) + tuple(z[1:])
                else:
                    d_pred, z_depth = self.G.decoders["d"].forward_nhwc(z)
            if want_s and "s" in self.opts.tasks:
                s_pred = self.G.decoders["s"].forward_nhwc(z, z_depth)

        def parts(t):
            return split_batch(t, G) if t is not None else [None] * G

        zs = list(zip(parts(z[0]), parts(z[1])))
        return zs, parts(d_pred), parts(z_depth), parts(s_pred)

    def get_masker_loss(self, multi_domain_batch):
        """reference trainer.py:1184-1254"""
        m_loss = 0
        doms = self._merged_masker_domains(multi_domain_batch)
        trunk = None
        if doms is not None and all(t in self.opts.tasks for t in "ds") and all(
                all(t in multi_domain_batch[d]["data"] for t in "ds") for d in doms):
            zs, d_preds, z_depths, s_preds = self._masker_trunk(multi_domain_batch, doms, want_s=True)
            trunk = {d: (zs[i], d_preds[i], z_depths[i], s_preds[i]) for i, d in enumerate(doms)}
        for domain, batch in multi_domain_batch.items():
            if domain == "rf":
                continue
            x = batch["data"]["x"]
            pre = trunk[domain] if trunk is not None else None
            z = pre[0] if pre is not None else self.G.encode(x)
            d_pred = s_pred = z_depth = None
            for task in ["d", "s", "m"]:
                if task not in batch["data"] or task not in self.opts.tasks:
                    continue
                target = batch["data"][task]
                if task == "d":
                    loss, d_pred, z_depth = self.masker_d_loss(x, z, target, domain, "G",
                                                               pre=None if pre is None else (pre[1], pre[2]))
                elif task == "s":
                    loss, s_pred = self.masker_s_loss(x, z, d_pred, z_depth, target, domain, "G",
                                                      pre=None if pre is None else pre[3])
                else:
                    cond = None
                    if self.opts.gen.m.use_spade:                                          # trainer.py:1233-1238
                        cond = self.G.make_m_cond(d_pred, s_pred, x)     # differentiable unless spade.detach
                    loss, _ = self.masker_m_loss(x, z, target, domain, "G", cond=cond, z_depth=z_depth,
                                                 depth_preds=d_pred)
                m_loss = m_loss + loss
        return m_loss

    def get_masker_d_loss(self, multi_domain_batch):
        """reference trainer.py:1109-1147 (Masker branch of get_D_loss): the generator side runs without a graph (the
        reference builds one and detaches the predictions, trainer.py:1113,1464,1578)."""
        total = 0
        adv = self.opts.train.lambdas.advent.adv_main
        doms = self._merged_masker_domains(multi_domain_batch)
        need_d = "d" in self.opts.tasks and (self.opts.gen.s.use_dada or self.opts.gen.m.use_dada or
                                             self.opts.gen.m.use_spade)
        trunk = None
        if doms is not None and need_d and all("s" in multi_domain_batch[d]["data"] for d in doms):
            with torch.no_grad():
                zs, d_preds, z_depths, s_preds = self._masker_trunk(multi_domain_batch, doms, want_s=True)
            trunk = {d: (zs[i], d_preds[i], z_depths[i], s_preds[i]) for i, d in enumerate(doms)}
        for domain, batch in multi_domain_batch.items():
            if domain == "rf":
                continue
            x = batch["data"]["x"]
            pre = trunk[domain] if trunk is not None else None
            with torch.no_grad():
                z = pre[0] if pre is not None else self.G.encode(x)
                d_pred = z_depth = s_pred = None
                if pre is not None:
                    d_pred, z_depth = pre[1], pre[2]
                elif "d" in self.opts.tasks and (self.opts.gen.s.use_dada or self.opts.gen.m.use_dada):
                    d_pred, z_depth = self.G.decoders["d"].forward_nhwc(z)
            if "s" in batch["data"] and "s" in self.opts.tasks:
                with torch.no_grad():
                    s_pred = pre[3] if pre is not None else self.G.decoders["s"].forward_nhwc(z, z_depth)
                if self.opts.gen.s.use_advent:                                             # trainer.py:1452
                    loss, _ = self._advent_d_term("s", s_pred, d_pred, domain)
                    total = total + loss * adv
            if "m" in batch["data"] and "m" in self.opts.tasks and self.opts.gen.m.use_advent:   # trainer.py:1563
                with torch.no_grad():
                    cond = None
                    if self.opts.gen.m.use_spade and "d" in self.opts.tasks:               # trainer.py:1127-1131
                        if d_pred is None:
                            d_pred, z_depth = self.G.decoders["d"].forward_nhwc(z)
                        if s_pred is None:
                            s_pred = self.G.decoders["s"].forward_nhwc(z, z_depth)
                        cond = self.G.make_m_cond(d_pred, s_pred, x)
                    logits = self.G.mask_nhwc(z, cond=cond, z_depth=z_depth)
                loss, _ = self._advent_d_term("m", logits, d_pred if self.opts.gen.m.use_dada else None, domain, x)
                total = total + loss * adv
        return total

    def _advent_d_term(self, task, pred, depth_preds, domain, x=None):
        w = self.opts.train.lambdas.advent.adv_main
        logits = ops.NHWC(pred.t.detach(), pred.c)
        dp = None
        if task == "s" and self.opts.gen.s.use_dada and depth_preds is not None:
            dp = ops.NHWC(depth_preds.t.detach(), depth_preds.c)
        if task == "m" and depth_preds is not None:                    # gen.m.use_dada, trainer.py:1566-1570
            dp = ops.resize_nearest(ops.detached(depth_preds), (x.shape[-2], x.shape[-1]))
        loss = self.losses["D"]["advent"](None, self.domain_labels[domain], self.D[task]["Advent"], dp, logits=logits,
                                          sigmoid_pair=task == "m") * w
        self.loss_log["D.%s.advent.%s" % (task, domain)] = loss.detach()
        return loss, None

    def _check_batch(self, multi_domain_batch):
        if not self.has_painter and "rf" in multi_domain_batch:
            raise ValueError("Trainer: an 'rf' batch needs the Painter task")
        if not self.has_masker and any(d != "rf" for d in multi_domain_batch):
            raise ValueError("Trainer: masker-domain batches need the masker tasks")

    def update_G(self, multi_domain_batch):
        """reference trainer.py:989-1015 + g_opt_step (674-683): D frozen, backward, extrapolate (even) / step (odd)."""
        self._check_batch(multi_domain_batch)
        for p in self.D.parameters():                                   # trainer.py:959-962
            p.requires_grad_(False)
        try:
            self.g_opt.zero_grad(set_to_none=True)
            g_loss = 0                                                  # get_G_loss, trainer.py:1162-1182
            do_m = self.has_masker and any(d != "rf" for d in multi_domain_batch)
            do_p = self.has_painter and "rf" in multi_domain_batch
            side = self._fork(self.G.compute_dtype) if (do_m and do_p and not self.use_pl4m) else None
            if side is not None:
                with torch.cuda.stream(side):
                    p_loss = self.get_painter_loss(multi_domain_batch)
                m_loss = self.get_masker_loss(multi_domain_batch)
                if not (isinstance(p_loss, torch.Tensor) and isinstance(m_loss, torch.Tensor)):
                    side = self._join(side)
                    g_loss = m_loss + p_loss
                else:
                    self._backward([m_loss, p_loss], self.G, side)
                    g_loss = m_loss.detach() + p_loss.detach()
            else:
                # The Painter's terms first, as on the forked path above: the label-noise draws of GANLoss come in the same
                # order whether or not the branches overlap (the reference shuffles the domain order per iteration,
                # trainer.py:948).  With pl4m the Masker's loss runs the Painter itself (one more power iteration of its
                # spectral norms): there the Masker domains go first, the order the pl4m fixtures were captured in.
                p_loss = self.get_painter_loss(multi_domain_batch) if (do_p and not self.use_pl4m) else 0
                if do_m:
                    g_loss = g_loss + self.get_masker_loss(multi_domain_batch)
                if do_p and self.use_pl4m:
                    p_loss = self.get_painter_loss(multi_domain_batch)
                g_loss = g_loss + p_loss
            if not isinstance(g_loss, torch.Tensor):
                # every term switched off (all lambdas 0): the reference would fail on ``int.backward()``; nothing to
                # differentiate and nothing for the optimizer to do
                return torch.zeros((), device=self.device)
            if side is None:
                self._backward(g_loss, self.G)
            if self.g_reducer is not None:
                self.g_reducer.finish()                                 # before extrapolation AND step (trainer.py:678-683)
            self._unscale_grads(self.G)
            if self.global_step % 2 == 0:
                self.g_opt.extrapolation()
            else:
                self.g_opt.step()
        finally:
            self._restore_d_grad_flags()                                # trainer.py:971-973
        return g_loss.detach()

    def _fork(self, dtype):
        """The side stream for one branch of an update, or None (overlap off / no device).  Everything both branches share
        is brought up to date on the calling stream first: the packed forms of the plain conv weights an optimizer step
        made stale (norms._PackCache: one batched launch that would otherwise be triggered by whichever branch gets there
        first, on ITS stream, while the other may be reading the buffers)."""
        if not self.overlap_branches or self.device.type != "cuda":
            return None
        from .norms import _PackCache
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
            for red in (self.g_reducer, self.d_reducer):           # gradients now come from two streams
                if red is not None:
                    red.streams = [torch.cuda.current_stream(self.device), self._side]
        _PackCache.repack_stale(dtype, self.device)
        self._side.wait_stream(torch.cuda.current_stream(self.device))
        if self.branch_events is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
            self.branch_events.append([e0, None, None])
        return self._side

    def _join(self, side):
        torch.cuda.current_stream(self.device).wait_stream(side)
        return None

    def _backward(self, loss, module, side=None):
        """``loss.backward()`` with the weight-gradient outputs of ``module`` carved from ONE zero-filled arena
        (ops.ZeroArena) instead of a fill launch per layer.  ``loss`` may be a list of independent losses whose graphs
        were recorded on two streams (``side``: the second one): autograd runs every node's backward on its forward's
        stream, so the two branches' kernels interleave on the device."""
        key = "_arena_numel_%d" % id(module)
        n = getattr(self, key, None)
        if n is None:
            n = sum((p.numel() + 63) // 64 * 64 + 64 for p in module.parameters() if p.requires_grad)
            setattr(self, key, n)
        dev = next(module.parameters()).device
        prev = ops.set_zero_arena(ops.ZeroArena(n, dev))
        try:
            ops.dgrad_prepack_run(side)      # every stride-1 data-gradient operator of this backward, one pack launch
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(dev))        # the arena's zeros, the packed operators
                torch.autograd.backward(list(loss))
                if self.branch_events is not None:
                    em, es = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    em.record(torch.cuda.current_stream(dev))
                    es.record(side)
                    self.branch_events[-1][1:] = [em, es]
                torch.cuda.current_stream(dev).wait_stream(side)        # the optimizer runs on the calling stream
            else:
                loss.backward()
        finally:
            ops.set_zero_arena(prev)

    @staticmethod
    def _unscale_grads(module):
        """fp16 loss scaling (autograd.set_grad_scale): every loss kernel wrote its input gradient times GRAD_SCALE, so
        the parameter gradients carry that factor; divide it out before the optimizer -- AFTER the reducer's
        ``finish()``: the bucket all-reduces were launched from hooks during the backward on the still-scaled
        gradients and ``finish()`` writes those averages back over ``p.grad`` (one fused multi-tensor multiply, nothing
        at the default scale of 1)."""
        from . import autograd as ag
        if ag.GRAD_SCALE != 1.0:
            grads = [p.grad for p in module.parameters() if p.grad is not None]
            if grads:
                torch._foreach_mul_(grads, 1.0 / ag.GRAD_SCALE)

    def _restore_d_grad_flags(self):
        for name, p in self.D.named_parameters():
            p.requires_grad_(not (name.endswith("weight_u") or name.endswith("weight_v")))

    def update_D(self, multi_domain_batch):
        """reference trainer.py:1017-1032 + d_opt_step (685-694)."""
        self._check_batch(multi_domain_batch)
        self.d_opt.zero_grad(set_to_none=True)
        d_loss = 0
        do_p = self.has_painter and "rf" in multi_domain_batch
        do_m = self.has_masker and any(d != "rf" for d in multi_domain_batch)
        side = self._fork(self.G.compute_dtype) if (do_m and do_p) else None
        if side is not None:
            with torch.cuda.stream(side):
                p_loss = self.get_D_loss(multi_domain_batch)
            m_loss = self.get_masker_d_loss(multi_domain_batch)
            if isinstance(m_loss, torch.Tensor) and isinstance(p_loss, torch.Tensor):
                self._backward([m_loss, p_loss], self.D, side)
                d_loss = m_loss.detach() + p_loss.detach()
            else:
                side = self._join(side)
                d_loss = m_loss + p_loss
        else:
            if do_p:
                d_loss = d_loss + self.get_D_loss(multi_domain_batch)
            if do_m:
                d_loss = d_loss + self.get_masker_d_loss(multi_domain_batch)
        if not isinstance(d_loss, torch.Tensor):     # no discriminator term is active (use_advent off, no Painter)
            return torch.zeros((), device=self.device)
        if side is None:
            self._backward(d_loss, self.D)
        if self.d_reducer is not None:
            self.d_reducer.finish()
        self._unscale_grads(self.D)
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

    def maybe_enable_pl4m(self):
        """The epoch check of the reference's ``train`` loop (trainer.py:899-909): from ``opts.gen.p.pl4m_epoch`` on, with a
        Painter of its own, task "p" and ``gen.m.use_pl4m``, the Painter's discriminator loss is back-propagated to the
        Masker.  ``run_epoch`` calls it; a host loop that drives ``train_step`` itself must call it once per epoch."""
        if (not self.use_pl4m and self.epoch == self.opts.gen.p.pl4m_epoch and "p" in self.opts.tasks
                and self.opts.gen.m.use_pl4m and sum(p.numel() for p in self.G.painter.parameters()) > 0):
            print("\n\n >>> Enabling pl4m at epoch {}\n\n".format(self.epoch))
            self.use_pl4m = True
        return self.use_pl4m

    def run_epoch(self, batches):
        """One epoch of the reference's loop (``train``'s pl4m check trainer.py:899-909, then ``run_epoch`` trainer.py:924-987)
        over an iterable of multi-domain batches (the zipped loaders are the caller's: data loading is out of scope):
        per batch ``update_G`` + ``update_D`` + step counter; per epoch the learning-rate schedulers and the epoch
        counter.  Returns the last (g_loss, d_loss)."""
        self.maybe_enable_pl4m()
        last = None
        for multi_domain_batch in batches:
            last = self.train_step(multi_domain_batch)
        self.update_learning_rates()                                    # trainer.py:983-984
        self.epoch += 1
        return last

    # ------------------------------------------------------------------------------------------ validation
    def eval_mode(self):
        """reference trainer.py:581-589"""
        if self.G is not None:
            self.G.eval()
        if self.D is not None:
            self.D.eval()
        self.current_mode = "eval"

    def train_mode(self):
        """reference trainer.py:591-600"""
        if self.G is not None:
            self.G.train()
        if self.D is not None:
            self.D.train()
        self.current_mode = "train"

    def get_G_loss(self, multi_domain_batch):
        """reference trainer.py:1162-1182: the Masker's and the Painter's generator-side terms of one multi-domain batch,
        no backward, no optimizer (``update_G`` = this + backward + the ExtraAdam half-step, on two streams); the logged
        terms land in ``loss_log``."""
        self._check_batch(multi_domain_batch)
        g_loss = 0
        if self.has_masker and any(d != "rf" for d in multi_domain_batch):
            g_loss = g_loss + self.get_masker_loss(multi_domain_batch)
        if self.has_painter and "rf" in multi_domain_batch:
            g_loss = g_loss + self.get_painter_loss(multi_domain_batch)
        return g_loss

    @torch.no_grad()
    def run_evaluation(self, val_batches, display_images=None):
        """reference trainer.py:1653-1704 over an iterable of multi-domain validation batches (the zipped loaders are the
        caller's): eval mode, ``get_G_loss`` per batch with the logged generator terms AVERAGED over the batches
        (sum_dict / div_dict, trainer.py:1674-1679), then ``eval_images("val", d)`` for d in r, s when the Masker has an
        m or s task, train mode again.  Comet image panels and the validation FID (logger.py, fid.py) are outside this
        path.  Returns {"losses": averaged terms, "metrics": {domain: eval_images' table}}."""
        self.eval_mode()
        sums, n = {}, 0
        for multi_domain_batch in val_batches:
            self.loss_log = {}
            self.get_G_loss(multi_domain_batch)
            for k, v in self.loss_log.items():
                if k.startswith("G."):
                    sums[k] = sums.get(k, 0.0) + float(v)
            n += 1
        losses = {k: v / max(n, 1) for k, v in sums.items()}
        metrics = {}
        if display_images is not None:
            self.display_images = display_images
        if ("m" in self.opts.tasks or "s" in self.opts.tasks) and getattr(self, "display_images", None):
            for d in ("r", "s"):
                self.eval_images("val", d)
                if ("val", d) in self.last_metrics:
                    metrics[d] = self.last_metrics[("val", d)]
        self.train_mode()
        return {"losses": losses, "metrics": metrics}

    @torch.no_grad()
    def eval_images(self, mode, domain):
        """reference trainer.py:1706-1799: accuracy and mIOU of the Masker's predictions over the display images
        ``self.display_images[mode][domain]`` (a list of ``{"data": {"x", "s", "m"[, "d"]}}`` samples, one image at a time
        as the reference does), per task m (binarised mask at 0.5; the 1-channel accuracy quirk of eval_metrics.py:67-76
        kept) and s (segmentation logits); the counts behind both metrics come from one HIP pass over the device tensors
        (``cgan_seg_counts``), nothing is moved to the host.  The averaged table is printed (the reference logs it to comet
        when an experiment exists), kept in ``self.last_metrics[(mode, domain)]``; returns 0 like the reference."""
        import numpy as np

        from . import functional as Fn
        from .eval_metrics import accuracy, mIOU
        from .utils import flatten_opts

        if domain == "s" and getattr(self, "kitti_pretrain", False):
            domain = "kitti"
        images = getattr(self, "display_images", None) or {}
        if domain == "rf" or domain not in images.get(mode, {}):
            return
        if not hasattr(self, "last_metrics"):
            self.last_metrics = {}
        metric_funcs = {"accuracy": accuracy, "mIOU": mIOU}
        scores = {"m": {}}
        if "s" in self.opts.tasks:
            scores["s"] = {}
        if "d" in self.opts.tasks and domain == "s" and self.opts.gen.d.classify.enable:
            raise NotImplementedError("eval_images: the depth CLASSIFICATION head (gen.d.classify.enable) is outside this build "
                                      "(SURVEY 2: depth regression only)")
        for task in scores:
            for key in metric_funcs:
                scores[task][key] = []
        for im_set in images[mode][domain]:
            data = im_set["data"]
            x = data["x"].unsqueeze(0).to(self.device)
            z = self.G.encode(x)
            z_depth = None
            if "s" in scores:
                if self.opts.gen.s.use_dada and "d" in self.opts.tasks:
                    _, z_depth = self.G.decoders["d"].forward_nhwc(z)
                s_pred = Fn.to_nchw(self.G.decoders["s"].forward_nhwc(z, z_depth))
                s = data["s"].unsqueeze(0).to(self.device)
                for name, fn in metric_funcs.items():
                    scores["s"][name].append(fn(s_pred, s))
            if "m" in self.opts.tasks:
                if z_depth is None and self.opts.gen.m.use_dada and "d" in self.opts.tasks:
                    _, z_depth = self.G.decoders["d"].forward_nhwc(z)
                # (cond stays None as in the reference, whose d_pred exists only with the classification head: G.mask builds
                # the SPADE conditioning itself when the mask decoder uses one, generator.py:257-262)
                pred_mask = (self.G.mask(x=x, z=z, cond=None, z_depth=z_depth) > 0.5).to(torch.float32)
                pred_prob = torch.cat([1 - pred_mask, pred_mask], dim=1)
                m = data["m"].unsqueeze(0).to(self.device)
                scores["m"]["accuracy"].append(accuracy(pred_mask, m))
                scores["m"]["mIOU"].append(mIOU(pred_prob, m))
        table = {}
        for task, met in scores.items():
            table[task] = {}
            for name, values in met.items():
                v = float(np.mean(values)) if values else float("nan")
                table[task][name] = v if not np.isnan(v) else -1
        self.last_metrics[(mode, domain)] = table
        print(f"metrics_{mode}_{domain}")
        print(flatten_opts(table))
        return 0

    # ------------------------------------------------------------------------------------------ checkpoints
    def update_learning_rates(self):
        """reference trainer.py:696-700 (called once per epoch by run_epoch, and epoch+1 times by resume)."""
        if self.g_scheduler is not None:
            self.g_scheduler.step()
        if self.d_scheduler is not None:
            self.d_scheduler.step()

    def save(self):
        """reference trainer.py:396-420: ``{epoch, G, g_opt, step[, D, d_opt]}`` to ``<output_path>/checkpoints/
        latest_ckpt.pth`` every call, plus ``epoch_<n>_ckpt.pth`` when ``epoch >= min_save_epoch`` and ``epoch %
        save_n_epochs == 0``.  Same keys, same state-dict layouts: the reference's ``resume`` reads these files."""
        from pathlib import Path

        save_dir = Path(self.opts.output_path) / "checkpoints"
        save_dir.mkdir(exist_ok=True, parents=True)   # the reference relies on output_path existing (train.py creates it)
        save_dict = {"epoch": self.epoch, "G": self.G.state_dict(), "g_opt": self.g_opt.state_dict(),
                     "step": self.global_step}
        if self.D is not None and sum(p.numel() for p in self.D.parameters()) > 0:
            save_dict["D"] = self.D.state_dict()
            save_dict["d_opt"] = self.d_opt.state_dict()
        if self.epoch >= self.opts.train.min_save_epoch and self.epoch % self.opts.train.save_n_epochs == 0:
            torch.save(save_dict, save_dir / ("epoch_%d_ckpt.pth" % self.epoch))
        torch.save(save_dict, save_dir / "latest_ckpt.pth")

    def _resolve_checkpoint(self):
        """The path rules of reference trainer.py:422-525: returns the checkpoint dict."""
        from pathlib import Path

        o = self.opts
        m_path, p_path, pm_path = Path(str(o.load_paths.m)), Path(str(o.load_paths.p)), Path(str(o.load_paths.pm))
        output_path = Path(o.output_path)

        def load(path):
            return torch.load(path, map_location=self.device, weights_only=False)

        def ckpt_file(path):
            if not path.exists():
                raise AssertionError("checkpoint path %s does not exist" % path)
            if path.is_dir():
                return path / "checkpoints/latest_ckpt.pth"
            if path.suffix != ".pth":
                raise AssertionError("checkpoint file %s is not a .pth" % path)
            return path

        if "m" in o.tasks and "p" in o.tasks:
            if all(str(p) == "none" for p in (m_path, p_path, pm_path)):
                return load(output_path / "checkpoints/latest_ckpt.pth")
            if str(pm_path) != "none":
                return load(ckpt_file(pm_path))
            if m_path != p_path:
                m_ckpt, p_ckpt = load(ckpt_file(m_path)), load(ckpt_file(p_path))
                return _merge(m_ckpt, p_ckpt)         # utils.merge(source=m, destination=p): m's entries win
            raise ValueError("Cannot resume a P+M model with provided load_paths:\n{}".format(o.load_paths))
        if str(m_path) != "none" and str(p_path) != "none":
            raise ValueError("Opts tasks are {} but received 2 values for the load_paths".format(o.tasks))
        if str(m_path) != "none":
            if "m" not in o.tasks:
                raise AssertionError("load_paths.m given but 'm' is not in opts.tasks")
            path = ckpt_file(m_path) if m_path.is_dir() else m_path
            if not m_path.exists():
                raise AssertionError("checkpoint path %s does not exist" % m_path)
            return load(path)
        if str(p_path) != "none":
            if "p" not in o.tasks:
                raise AssertionError("load_paths.p given but 'p' is not in opts.tasks")
            if not p_path.exists():
                raise AssertionError("checkpoint path %s does not exist" % p_path)
            return load(p_path / "checkpoints/latest_ckpt.pth" if p_path.is_dir() else p_path)
        return load(output_path / "checkpoints/latest_ckpt.pth")

    def resume(self, inference=False):
        """reference trainer.py:422-579: load G (``strict=False`` with warnings in inference mode, then stop), g_opt,
        replay the schedulers ``epoch + 1`` times, D and d_opt, epoch / step, and round the step up to an even number
        (extragradient: extrapolation happens on even steps)."""
        checkpoint = self._resolve_checkpoint()
        if inference:
            bad = self.G.load_state_dict(checkpoint["G"], strict=False)
            if bad.missing_keys:
                print("WARNING: Missing keys in self.G.load_state_dict, keeping inits")
                print(bad.missing_keys)
            if bad.unexpected_keys:
                print("WARNING: Ignoring Unexpected keys in self.G.load_state_dict")
                print(bad.unexpected_keys)
            return
        self.G.load_state_dict(checkpoint["G"])
        self.g_opt.load_state_dict(checkpoint["g_opt"])
        for _ in range(self.epoch + 1):              # trainer.py:557-558 (self.logger.epoch is still the pre-resume value)
            self.update_learning_rates()
        if self.D is not None and sum(p.numel() for p in self.D.parameters()) > 0:
            self.D.load_state_dict(checkpoint["D"])
            self.d_opt.load_state_dict(checkpoint["d_opt"])
        self.epoch = checkpoint["epoch"]
        self.global_step = checkpoint["step"]
        if self.global_step % 2 != 0:
            self.global_step += 1

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
    def _resize_input(self, x, size):
        """F.interpolate(x, size, mode="bilinear") of an NCHW batch on the 16-bit NHWC representation -> fp32 NCHW."""
        x16 = ops.resize_bilinear(ops.nchw_to_nhwc(x.float(), self.G.compute_dtype), size, align_corners=False)
        return ops.nhwc_to_nchw(x16)

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
        x = x.half() if half else x.float()
        x = x.contiguous()
        if auto_resize_640 and (x.shape[-1] != 640 or x.shape[-2] != 640):
            # trainer.py:259-261: F.interpolate(x, (640, 640), mode="bilinear").  The Masker reads x as 16-bit NHWC
            # anyway, so the resize runs on that representation (same kernel as the decoders' bilinear resizes).
            x = self._resize_input(x, (640, 640))
            x = x.half() if half else x.float()

        self.G.painter.set_latent_shape(x.shape, True)                                   # trainer.py:266

        def timed(key):
            # the reference brackets every stage with a device-synchronising Timer (trainer.py:268-310); here a stage is
            # only synchronised when the caller asked for its time (a key of ``stores``): 16 device drains per batch less
            return Timer(store=stores.get(key), ignore=key not in stores)

        with timed("all events"):
            with timed("encode"):
                z = self.G.encode(x)
            with timed("depth"):
                depth_nhwc, z_depth = self.G.decoders["d"].forward_nhwc(z)
            with timed("segmentation"):
                seg_nhwc = self.G.decoders["s"].forward_nhwc(z, z_depth)
            cond = None
            if isinstance(seg_nhwc, ops.PairMap):
                # split-precision Masker (G.set_compute_dtype("split24" | "pair16")): the mask decoder stays on the split maps
                # (the mask leaves as fp32), its conditioning map included (SPADE mask decoder: built from the split depth /
                # segmentation maps in fp32); the event kernels read the maps rounded once to 16 bit
                if self.opts.gen.m.use_spade:
                    cond = self.G.make_m_cond(depth_nhwc, seg_nhwc, x)                                        # :285
                depth_nhwc, seg_nhwc = ops.pair_to_nhwc(depth_nhwc), ops.pair_to_nhwc(seg_nhwc)
            elif self.opts.gen.m.use_spade:
                cond = self.G.make_m_cond(depth_nhwc, seg_nhwc, x)                                            # :285
            with timed("mask"):
                mask = self.G.mask(z=z, cond=cond, z_depth=z_depth)
                mask = mask if isinstance(z[0], ops.PairMap) else mask.to(x.dtype)     # pair16: the fp32 mask is binarised

            wildfire = smog = flood = None
            # the flood painter is independent of the other two events: it runs on the side stream beside them (only when
            # no per-event timing was asked for)
            side = None
            if "flood" not in ignore_event and not stores and self.overlap_branches and self.device.type == "cuda":
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                side = self._side
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):
                    flood = self.compute_flood(x, m=mask, s=seg_nhwc, cloudy=cloudy, bin_value=bin_value)
            if "wildfire" not in ignore_event:
                with timed("wildfire"):
                    wildfire = self.compute_fire(x, seg_preds=seg_nhwc)
            if "smog" not in ignore_event:
                with timed("smog"):
                    smog = self.compute_smog(x, d=depth_nhwc, s=seg_nhwc)
            if "flood" not in ignore_event and side is None:
                with timed("flood"):
                    flood = self.compute_flood(x, m=mask, s=seg_nhwc, cloudy=cloudy, bin_value=bin_value)
            if side is not None:
                torch.cuda.current_stream(self.device).wait_stream(side)

        output_data = {}
        with Timer(store=stores.get("numpy"), ignore=not numpy or "numpy" not in stores):
            for name, ev in (("flood", flood), ("wildfire", wildfire), ("smog", smog)):
                if ev is None:
                    continue
                if numpy:
                    ev = ops.normalize_to_uint8(ev).cpu().numpy()                        # trainer.py:311-326
                output_data[name] = ev
        if return_masks:
            output_data["mask"] = ops.binarize(mask, bin_value, want_float=False, want_uint8=True).cpu().numpy()
        return output_data
