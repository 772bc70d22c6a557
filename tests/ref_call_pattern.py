"""The reference trainer's use of the model API, restated for the boundary test.

``/root/reference/climategan/trainer.py`` cannot travel to the GPU box, so this file plays its part: it drives a
``climategan_amd`` generator / discriminator / loss dictionary through ONE G update and ONE D update using only the
calls, argument types and return types the reference's own ``get_masker_loss`` / ``masker_{d,s,m}_loss`` /
``get_painter_loss`` / ``get_D_loss`` / ``painter_loss_for_masker`` make (trainer.py:1034-1160, 1184-1254, 1256-1387,
1389-1651):

* ``G.encode(x)`` -> an opaque latent handed back to ``G.decoders[t](z, ...)``; ``z[0].shape[0]`` is read (trainer.py:602-607);
* ``G.decoders["d"](z) -> (NCHW depth, z_depth)``, ``G.decoders["s"](z, z_depth) -> NCHW logits``,
  ``G.decoders["m"](z, cond=, z_depth=) -> NCHW logits``, ``G.make_m_cond(d, s, x)`` on those NCHW tensors;
* torch expressions on the results (``softmax``, ``sigmoid``, ``1 - p``, ``cat``, ``detach``, ``fake * m``, ``interpolate``);
* ``G.paint(m, x) -> NCHW``, ``D["p"](NCHW batch concatenation) -> list[num_D] of list[n_layers + 2] NCHW tensors``,
  ``divide_pred``, ``vgg_preprocess``; ``D[t]["Advent"]`` through ``ADVENTAdversarialLoss(prob, label, D, depth)``;
* the loss objects of ``get_losses`` called with NCHW tensors; ``loss.backward()``; ``requires_grad`` toggling.

Nothing here touches ``ops.NHWC``, ``forward_nhwc``, ``nhwc=True`` or ``climategan_amd.autograd``.  Terms are logged
under the keys ``climategan_amd.trainer.Trainer.loss_log`` uses so the same assertions read both."""
import torch

from climategan_amd.tutils import divide_pred, vgg_preprocess


class ReferenceCalls:
    def __init__(self, trainer):
        """``trainer``: a set-up ``climategan_amd.trainer.Trainer`` -- used as the holder of G, D, losses, the two
        optimizers and opts (what the reference's ``Trainer.setup`` builds); none of ITS loss code is called."""
        self.T = trainer
        self.G, self.D, self.losses, self.opts = trainer.G, trainer.D, trainer.losses, trainer.opts
        self.labels = {"s": 0, "r": 1}
        self.log = trainer.loss_log
        self.use_pl4m = False

    # ------------------------------------------------------------------ generator side
    def depth_term(self, z, target, domain):
        w = self.opts.train.lambdas.G.d.main
        pred, z_depth = self.G.decoders["d"](z)
        assert pred.dim() == 4 and pred.dtype == torch.float32          # an NCHW tensor, as in the reference
        loss = self.losses["G"]["tasks"]["d"](pred, target) * w
        if w == 0 or domain == "r":
            return torch.zeros((), device=pred.device), pred, z_depth
        self.log["G.d." + domain] = loss.detach()
        return loss, pred, z_depth

    def seg_term(self, z, d_pred, z_depth, target, domain, side):
        o, lam = self.opts, self.opts.train.lambdas
        total = torch.zeros((), device=self.T.device)
        pred = self.G.decoders["s"](z, z_depth) if (side == "G" or o.gen.s.use_advent) else None
        probs = None
        if side == "G":
            if domain == "s" and lam.G["s"]["crossent"] != 0:
                term = self.losses["G"]["tasks"]["s"]["crossent"](pred, target.squeeze(1)) * lam.G["s"]["crossent"]
                self.log["G.s.crossent.s"] = term.detach()
                total = total + term
            if domain == "r" and lam.G["s"]["minent"] != 0:
                probs = torch.softmax(pred, dim=1)
                term = self.losses["G"]["tasks"]["s"]["minent"](probs) * lam.G["s"]["minent"]
                self.log["G.s.minent.r"] = term.detach()
                total = total + term
        if o.gen.s.use_advent:
            depth = d_pred.detach() if (o.gen.s.use_dada and d_pred is not None) else None
            if side == "D":
                label, fn, w, pred = domain, self.losses["D"]["advent"], lam.advent.adv_main, pred.detach()
            else:
                label, fn, w = "s", self.losses["G"]["tasks"]["s"]["advent"], lam.G["s"]["advent"]
            if (side == "D" or domain == "r") and w != 0:
                if probs is None or side == "D":
                    probs = torch.softmax(pred, dim=1)
                term = fn(probs, self.labels[label], self.D["s"]["Advent"], depth) * w
                self.log["%s.s.advent.%s" % (side, domain)] = term.detach()
                total = total + term
        return total, pred

    def mask_term(self, x, z, target, domain, side, cond=None, z_depth=None, d_pred=None):
        o, lam = self.opts, self.opts.train.lambdas
        total = torch.zeros((), device=self.T.device)
        logits = self.G.decoders["m"](z, cond=cond, z_depth=z_depth)
        assert logits.dim() == 4 and logits.shape[1] == 1
        p = torch.sigmoid(logits)
        prob = torch.cat([p, 1 - p], dim=1)
        if side == "G":
            terms = []
            if lam.G.m.tv != 0:
                terms.append(("tv." + domain, self.losses["G"]["tasks"]["m"]["tv"](p) * lam.G.m.tv))
            if domain == "s" and lam.G.m.bce != 0:
                terms.append(("bce.s", self.losses["G"]["tasks"]["m"]["bce"](logits, target) * lam.G.m.bce))
            if domain == "r":
                if o.gen.m.use_ground_intersection and lam.G["m"]["gi"] != 0:
                    terms.append(("gi.r", self.losses["G"]["tasks"]["m"]["gi"](p, target) * lam.G["m"]["gi"]))
                if self.use_pl4m and lam.G.m.pl4m != 0:
                    terms.append(("pl4m.r", self.painter_for_masker(x, p) * lam.G.m.pl4m))
                if o.gen.m.use_minent and lam.advent.ent_main != 0:
                    terms.append(("minent.r", self.losses["G"]["tasks"]["m"]["minent"](prob) * lam.advent.ent_main))
            for key, term in terms:
                self.log["G.m." + key] = term.detach()
                total = total + term
        if o.gen.m.use_advent:
            depth = None
            if o.gen.m.use_dada and d_pred is not None:
                depth = torch.nn.functional.interpolate(d_pred.detach(), size=x.shape[-2:], mode="nearest")
            if side == "D":
                label, fn, prob = domain, self.losses["D"]["advent"], prob.detach()
            else:
                label, fn = "s", self.losses["G"]["tasks"]["m"]["advent"]
            w = lam.advent.adv_main
            if (side == "D" or domain == "r") and w != 0:
                term = fn(prob.to(self.T.device), self.labels[label], self.D["m"]["Advent"], depth) * w
                self.log["%s.m.advent.%s" % (side, domain)] = term.detach()
                total = total + term
        return total, prob

    def painter_for_masker(self, x, m):
        # (the reference re-enables EVERY Painter parameter afterwards, its spectral-norm u / v included -- a quirk the
        # package does not copy, see tests/test_gpu_configs_640.py::_compare_grads; here: the ones that were trainable)
        trainable = [p for p in self.G.painter.parameters() if p.requires_grad]
        for p in trainable:
            p.requires_grad = False
        fake = self.G.paint(m, x)
        both = torch.cat([torch.cat([m, x], axis=1), torch.cat([m, fake], axis=1)], dim=0)
        _, fake_d = divide_pred(self.D["p"](both))
        loss = self.losses["G"]["p"]["gan"](fake_d, True, False)
        if "p" in self.opts.tasks:
            for p in trainable:
                p.requires_grad = True
        return loss

    def masker_loss(self, batches):
        total = 0
        for domain, batch in batches.items():
            if domain == "rf":
                continue
            data = batch["data"]
            x = data["x"]
            z = self.G.encode(x)
            assert x.shape[0] == (z[0].shape[0] if isinstance(z, (list, tuple)) else z.shape[0])
            d_pred = s_pred = z_depth = None
            for task in ("d", "s", "m"):
                if task not in data:
                    continue
                if task == "d":
                    term, d_pred, z_depth = self.depth_term(z, data["d"], domain)
                elif task == "s":
                    term, s_pred = self.seg_term(z, d_pred, z_depth, data["s"], domain, "G")
                else:
                    cond = None
                    if self.opts.gen.m.use_spade:
                        if not self.opts.gen.m.detach:
                            d_pred, s_pred = d_pred.clone(), s_pred.clone()
                        cond = self.G.make_m_cond(d_pred, s_pred, x)
                    term, _ = self.mask_term(x, z, data["m"], domain, "G", cond=cond, z_depth=z_depth, d_pred=d_pred)
                total = total + term
        return total

    def painter_loss(self, batches):
        lam = self.opts.train.lambdas.G.p
        data = batches["rf"]["data"]
        x, m = data["x"], data["m"]
        fake = self.G.paint(m, x)
        assert fake.shape == x.shape and fake.dtype == x.dtype and fake.requires_grad
        total = 0
        if lam.vgg != 0:
            term = self.losses["G"]["p"]["vgg"](vgg_preprocess(fake * m), vgg_preprocess(x * m)) * lam.vgg
            self.log["G.p.vgg"] = term.detach()
            total = total + term
        both = torch.cat([torch.cat([m, x], axis=1), torch.cat([m, fake], axis=1)], dim=0)
        out = self.D["p"](both)
        assert isinstance(out, list) and isinstance(out[0], list) and out[0][0].dim() == 4
        real_d, fake_d = divide_pred(out)
        term = self.losses["G"]["p"]["gan"](fake_d, True, False)
        self.log["G.p.gan"] = term.detach()
        total = total + term
        if self.opts.dis.p.get_intermediate_features and lam.featmatch != 0:
            term = self.losses["G"]["p"]["featmatch"](real_d, fake_d) * lam.featmatch
            self.log["G.p.featmatch"] = term.detach()
            total = total + term
        return total

    def update_G(self, batches, step=0):
        for p in self.D.parameters():
            p.requires_grad = False
        self.T.g_opt.zero_grad()
        loss = self.masker_loss(batches) + self.painter_loss(batches)
        loss.backward()
        (self.T.g_opt.extrapolation if step % 2 == 0 else self.T.g_opt.step)()
        for name, p in self.D.named_parameters():
            p.requires_grad = not name.endswith(("weight_u", "weight_v"))
        return loss.detach()

    # ------------------------------------------------------------------ discriminator side
    def d_loss(self, batches):
        total = 0
        adv = self.opts.train.lambdas.advent.adv_main
        for domain, batch in batches.items():
            data = batch["data"]
            x = data["x"]
            if domain == "rf":
                m = data["m"]
                with torch.no_grad():
                    fake = self.G.paint(m, x)
                    fake = fake.detach()
                    fake.requires_grad_()
                both = torch.cat([torch.cat([m, x], axis=1), torch.cat([m, fake], axis=1)], dim=0)
                real_d, fake_d = divide_pred(self.D["p"](both))
                term = self.losses["D"]["p"](fake_d, False, True)
                term = term + self.losses["D"]["p"](real_d, True, True)
                self.log["D.p.gan"] = term.detach()
                total = total + term
                continue
            z = self.G.encode(x)
            s_pred = d_pred = cond = z_depth = None
            if "s" in data:
                if "d" in self.opts.tasks and self.opts.gen.s.use_dada:
                    d_pred, z_depth = self.G.decoders["d"](z)
                term, s_pred = self.seg_term(z, d_pred, z_depth, None, domain, "D")
                total = total + term * adv
            if "m" in data:
                if "d" in self.opts.tasks:
                    if self.opts.gen.m.use_spade:
                        if d_pred is None:
                            d_pred, z_depth = self.G.decoders["d"](z)
                        cond = self.G.make_m_cond(d_pred, s_pred, x)
                    elif self.opts.gen.m.use_dada and d_pred is None:
                        d_pred, z_depth = self.G.decoders["d"](z)
                term, _ = self.mask_term(x, z, None, domain, "D", cond=cond, z_depth=z_depth, d_pred=d_pred)
                total = total + term * adv
        return total

    def update_D(self, batches, step=0):
        self.T.d_opt.zero_grad()
        loss = self.d_loss(batches)
        loss.backward()
        (self.T.d_opt.extrapolation if step % 2 == 0 else self.T.d_opt.step)()
        return loss.detach()
