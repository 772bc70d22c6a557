"""GPU parity of the training path: the Painter discriminator update (forward, GANLoss, backward through spectral-norm
convs / instance norm / LeakyReLU, all HIP) against the loss and gradients captured from the REAL reference
(oracle/make_golden.py: dstep_p), then one ExtraAdam extrapolation + step on those gradients.

Tolerance.  Activations AND activation gradients are stored in 16 bit between kernels while the reference runs in
fp32.  The GAN-loss gradient has one sign per half-batch, so every instance-norm backward subtracts a large common
mode (``dz - mean(dz)``) from 16-bit values and keeps their rounding error: measured on MI355X, the last conv's
gradient (no norm behind it) is within 4e-4 relative L2 of the reference, one instance-norm further back 1-3 %, the
first layers 4 % in fp16 (bf16: 1.6 % ... 14 %); loss scaling does not change this (it is cancellation, not
underflow).  Bound enforced per parameter tensor: relative L2 error <= 8e-2 and cosine >= 0.997 in fp16 (0.25 / 0.96
in bf16), bias gradients in front of a norm (exactly zero in the reference) below 1e-3 (bf16 8e-3) of the layer's
weight-gradient scale; loss within 2e-3 relative (bf16: 1e-2)."""
import numpy as np
import pytest
import torch

from helpers import case_state_dict, golden_cases, load_golden, t
from oracle.make_golden import case_inputs

pytestmark = pytest.mark.gpu
NAME = "dstep_p"
BOUNDS = {torch.float16: (8e-2, 0.997, 2e-3), torch.bfloat16: (0.25, 0.96, 1e-2)}


def build_D(case, dt):
    from climategan_amd.discriminator import define_D

    D = define_D(input_nc=4, ndf=case["ndf"], n_layers=case["n_layers"], norm="instance", use_sigmoid=False,
                 get_intermediate_features=True, num_D=case["num_D"]).cuda()
    D.load_state_dict(case_state_dict(case), strict=True)
    D.compute_dtype = dt
    D.train()
    return D


def d_loss(D, inp):
    from climategan_amd.losses import GANLoss
    from climategan_amd.tutils import divide_pred

    gan = GANLoss(use_lsgan=False, soft_shift=0.0, flip_prob=0.0)
    real_cat = torch.cat([inp["m"], inp["x"]], dim=1)
    fake_cat = torch.cat([inp["m"], inp["fake"]], dim=1)
    real_fake_d = D(torch.cat([real_cat, fake_cat], dim=0), nhwc=True)
    real_d, fake_d = divide_pred(real_fake_d)
    loss = gan(fake_d, False, True)
    loss = loss + gan(real_d, True, True)
    return loss


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_painter_d_step_matches_reference(dt):
    case = golden_cases()[NAME]
    gold = load_golden(NAME)
    D = build_D(case, dt)
    inp = {k: t(v).cuda() for k, v in case_inputs(NAME, case).items()}
    loss = d_loss(D, inp)
    loss.backward()
    l2_max, cos_min, loss_tol = BOUNDS[dt]
    assert abs(loss.item() - float(gold["loss"][0])) <= loss_tol * abs(float(gold["loss"][0]))
    checked, bad = 0, []
    for key, p in D.named_parameters():
        if not p.requires_grad:
            assert p.grad is None
            continue
        ref = gold["grad." + key].astype(np.float64)
        got = p.grad.cpu().numpy().astype(np.float64)
        assert got.shape == ref.shape and p.grad.dtype == torch.float32, key
        wscale = np.abs(gold["grad." + key.rsplit(".", 1)[0] + ".weight_bar"]).max()
        if key.endswith("bias") and np.abs(ref).max() < 1e-6 * wscale:
            ok = np.abs(got).max() <= (1e-3 if dt == torch.float16 else 8e-3) * wscale   # exactly zero in the reference
            stat = ("zero-bias", np.abs(got).max() / wscale)
        else:
            l2 = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
            cos = (got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum())
            ok = l2 <= l2_max and cos >= cos_min
            stat = (l2, cos)
        if not ok:
            bad.append((key,) + stat)
        checked += 1
    assert not bad, bad
    assert checked == 2 * 5 * case["num_D"]
    sd = D.state_dict()
    for k in gold:
        if k.startswith("post."):
            assert np.abs(sd[k[5:]].cpu().numpy() - gold[k]).max() < 2e-5, k


def test_d_update_with_extra_adam_moves_parameters():
    """update_D (trainer.py:1017-1032) end to end on the HIP path: loss.backward() then ExtraAdam extrapolation on an
    even step and step on an odd one (trainer.py:685-694); the loss on the same batch goes down."""
    from climategan_amd.optim import ExtraAdam

    case = golden_cases()[NAME]
    D = build_D(case, torch.float16)
    inp = {k: t(v).cuda() for k, v in case_inputs(NAME, case).items()}
    opt = ExtraAdam([p for p in D.parameters() if p.requires_grad], lr=2e-3, betas=(0.5, 0.999))
    losses = []
    for step in range(4):
        opt.zero_grad()
        loss = d_loss(D, inp)
        loss.backward()
        losses.append(loss.item())
        if step % 2 == 0:
            opt.extrapolation()
        else:
            opt.step()
    assert losses[-1] < losses[0], losses


def test_nchw_outputs_carry_a_graph_under_autograd():
    """``D(x)`` with the reference's signature (NCHW in, list of lists of NCHW tensors out, discriminator.py:172-182,
    227-239): under autograd the tensors carry their graph back to the parameters AND to an NCHW input that wants a
    gradient (the reference's ``fake.requires_grad_()``, trainer.py:1085); values equal the NHWC maps'."""
    from climategan_amd import ops
    case = golden_cases()[NAME]
    D = build_D(case, torch.float16)
    sd = {k: v.clone() for k, v in D.state_dict().items()}
    inp = {k: t(v).cuda() for k, v in case_inputs(NAME, case).items()}
    x = torch.cat([inp["m"], inp["x"]], dim=1).requires_grad_(True)
    with torch.no_grad():
        ref = D(x.detach(), nhwc=True)
    D.load_state_dict(sd)                                 # the spectral-norm u / v of the first call
    out = D(x)
    assert len(out) == case["num_D"] and out[0][0].shape[1] == case["ndf"] and out[0][-1].requires_grad
    for a, b in zip(out, ref):
        for ta, tb in zip(a, b):
            assert torch.equal(ta.detach(), ops.nhwc_to_nchw(tb))
    sum(o[-1].mean() for o in out).backward()
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().max() > 0
    missing = [k for k, p in D.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing


# ------------------------------------------------------------------------------------------------ G side
GNAME = "gstep_p"


def build_trainer(case, dt, vgg=False):
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer
    from helpers import gstep_d_state_dict

    opts = default_opts()
    opts.tasks = ["p"]
    opts.gen.p.latent_dim = case["latent_dim"]
    opts.gen.p.spade_n_up = case["n_up"]
    opts.dis.p.ndf, opts.dis.p.n_layers, opts.dis.p.num_D = case["ndf"], case["n_layers"], case["num_D"]
    opts.dis.soft_shift, opts.dis.flip_prob = 0.0, 0.0
    if not vgg:
        opts.train.lambdas.G.p.vgg = 0
    T = Trainer(opts, device="cuda").setup(inference=False)
    T.G.painter.load_state_dict(case_state_dict(case), strict=True)
    T.D["p"].load_state_dict(gstep_d_state_dict(case), strict=True)
    T.G.set_compute_dtype(dt)
    T.D.set_compute_dtype(dt)
    T.G.painter.set_latent_shape((case["B"], 3, case["H"], case["W"]), True)
    return T


def test_painter_g_step_matches_reference():
    """get_painter_loss + backward on the HIP path vs the reference: loss terms, and the gradient of every trainable
    Painter tensor (through the frozen D, the paste, 10 SPADE blocks, spectral norm).  Same 16-bit caveats as the D
    step, over a much deeper graph: per tensor, cosine >= 0.99 and relative L2 <= 0.15 for the weight tensors; bias
    gradients in front of an instance norm (zero in the reference) below 1e-2 of their layer's weight-gradient scale."""
    case = golden_cases()[GNAME]
    gold = load_golden(GNAME)
    T = build_trainer(case, torch.float16)
    inp = {k: t(v).cuda() for k, v in case_inputs(GNAME, case).items()}
    batch = {"rf": {"data": {"x": inp["x"], "m": inp["m"]}}}
    for p in T.D.parameters():
        p.requires_grad_(False)
    loss = T.get_painter_loss(batch)
    loss.backward()
    assert abs(T.loss_log["G.p.gan"].item() - float(gold["gan"][0])) <= 5e-3 * abs(float(gold["gan"][0]))
    assert abs(T.loss_log["G.p.featmatch"].item() - float(gold["featmatch"][0])) <= 1e-2 * abs(float(gold["featmatch"][0]))
    bad, checked = [], 0
    for key, p in T.G.painter.named_parameters():
        if not p.requires_grad:
            continue
        assert p.grad is not None, key
        ref = gold["grad." + key].astype(np.float64)
        got = p.grad.cpu().numpy().astype(np.float64)
        base = key.rsplit(".", 1)[0]
        wkey = "grad." + base + (".weight_bar" if "grad." + base + ".weight_bar" in gold else ".weight")
        wscale = np.abs(gold[wkey]).max()
        if key.endswith("bias") and np.abs(ref).max() < 1e-4 * wscale:
            if np.abs(got).max() > 1e-2 * wscale:
                bad.append((key, "zero-bias", np.abs(got).max() / wscale))
        else:
            l2 = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
            cos = (got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum())
            if not (l2 <= 0.15 and cos >= 0.99):
                bad.append((key, l2, cos))
        checked += 1
    assert not bad, bad
    assert checked == sum(1 for k in gold if k.startswith("grad."))


def test_painter_g_step_optional_terms_and_lsgan_match_reference():
    """The non-default branches of get_painter_loss (trainer.py:1289-1315): TVLoss(fake_flooded * m), ContextLoss,
    ReconstructionLoss with non-zero lambdas, and the LSGAN form of GANLoss (losses.py:50-52), against the reference's own
    classes on a soft mask (golden ``gstep_p_aux``): every term's value, and the gradient of every trainable Painter tensor
    (the fused image-space kernel's gradient joins the heads' gradient in front of the Painter)."""
    from climategan_amd.losses import GANLoss

    name = "gstep_p_aux"
    case = golden_cases()[name]
    gold = load_golden(name)
    T = build_trainer(case, torch.float16)
    lam = T.opts.train.lambdas.G.p
    lam.tv, lam.context, lam.reconstruction = case["aux"]["tv"], case["aux"]["context"], case["aux"]["reconstruction"]
    T.losses["G"]["p"]["gan"] = GANLoss(use_lsgan=True, soft_shift=0.0, flip_prob=0.0)
    inp = {k: t(v).cuda() for k, v in case_inputs(name, case).items()}
    batch = {"rf": {"data": {"x": inp["x"], "m": inp["m"]}}}
    for p in T.D.parameters():
        p.requires_grad_(False)
    loss = T.get_painter_loss(batch)
    loss.backward()
    for key, log, tol in (("gan", "G.p.gan", 5e-3), ("featmatch", "G.p.featmatch", 1e-2), ("tv", "G.p.tv", 1e-2),
                          ("context", "G.p.context", 5e-3), ("reconstruction", "G.p.reconstruction", 5e-3)):
        ref, got = float(gold[key][0]), T.loss_log[log].item()
        assert abs(got - ref) <= tol * abs(ref), (key, got, ref)
    assert abs(loss.item() - float(gold["loss"][0])) <= 1e-2 * float(gold["loss"][0])
    bad, checked = [], 0
    for key, p in T.G.painter.named_parameters():
        if not p.requires_grad:
            continue
        ref = gold["grad." + key].astype(np.float64)
        got = p.grad.cpu().numpy().astype(np.float64)
        base = key.rsplit(".", 1)[0]
        wkey = "grad." + base + (".weight_bar" if "grad." + base + ".weight_bar" in gold else ".weight")
        wscale = np.abs(gold[wkey]).max()
        if key.endswith("bias") and np.abs(ref).max() < 1e-4 * wscale:
            if np.abs(got).max() > 1e-2 * wscale:
                bad.append((key, "zero-bias", np.abs(got).max() / wscale))
        else:
            l2 = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
            cos = (got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum())
            if not (l2 <= 0.15 and cos >= 0.99):
                bad.append((key, l2, cos))
        checked += 1
    assert not bad, bad
    assert checked == sum(1 for k in gold if k.startswith("grad."))


def test_painter_local_pair_with_image_space_terms_matches_reference():
    """Both non-default branches of get_painter_loss together (round 5; the combination used to raise): TV / context /
    reconstruction (trainer.py:1289-1315) in front of the local / global pair (:1323-1358) on a soft mask, against the
    reference's own classes (golden ``gstep_p_local_aux``): every term and the gradient of every trainable Painter tensor."""
    name = "gstep_p_local_aux"
    case = golden_cases()[name]
    gold = load_golden(name)
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer
    from helpers import disc_p_shapes
    from climategan_amd import fill

    opts = default_opts()
    opts.tasks = ["p"]
    opts.gen.p.latent_dim, opts.gen.p.spade_n_up = case["latent_dim"], case["n_up"]
    opts.dis.p.ndf, opts.dis.p.n_layers, opts.dis.p.num_D = case["ndf"], case["n_layers"], case["num_D"]
    opts.dis.p.use_local_discriminator = True
    opts.dis.soft_shift, opts.dis.flip_prob = 0.0, 0.0
    lam = opts.train.lambdas.G.p
    lam.vgg, lam.gan = 0, case["local"]["lambda_gan"]
    lam.tv, lam.context, lam.reconstruction = case["aux"]["tv"], case["aux"]["context"], case["aux"]["reconstruction"]
    T = Trainer(opts, device="cuda").setup(inference=False)
    T.G.painter.load_state_dict(case_state_dict(case), strict=True)
    shapes = disc_p_shapes(3, case["ndf"], case["n_layers"], case["num_D"])
    for i, which in enumerate(("global", "local")):
        T.D["p"][which].load_state_dict({k: t(v) for k, v in fill.fill_state_dict(shapes, case["seed"] + 1 + i).items()},
                                        strict=True)
    T.G.set_compute_dtype(torch.float16)
    T.D.set_compute_dtype(torch.float16)
    T.G.painter.set_latent_shape((case["B"], 3, case["H"], case["W"]), True)
    inp = {k: t(v).cuda() for k, v in case_inputs(name, case).items()}
    batch = {"rf": {"data": {"x": inp["x"], "m": inp["m"]}}}
    for p in T.D.parameters():
        p.requires_grad_(False)
    loss = T.get_painter_loss(batch)
    loss.backward()
    for key, log, tol in (("gan", "G.p.gan", 5e-3), ("featmatch", "G.p.featmatch", 1e-2), ("tv", "G.p.tv", 1e-2),
                          ("context", "G.p.context", 5e-3), ("reconstruction", "G.p.reconstruction", 5e-3)):
        ref, got = float(gold[key][0]), float(T.loss_log[log])
        assert abs(got - ref) <= tol * abs(ref), (key, got, ref)
    assert abs(loss.item() - float(gold["loss"][0])) <= 1e-2 * float(gold["loss"][0])
    bad, checked = [], 0
    for key, p in T.G.painter.named_parameters():
        if not p.requires_grad:
            continue
        ref = gold["grad." + key].astype(np.float64)
        got = p.grad.cpu().numpy().astype(np.float64)
        base = key.rsplit(".", 1)[0]
        wkey = "grad." + base + (".weight_bar" if "grad." + base + ".weight_bar" in gold else ".weight")
        wscale = np.abs(gold[wkey]).max()
        if key.endswith("bias") and np.abs(ref).max() < 1e-4 * wscale:
            if np.abs(got).max() > 1e-2 * wscale:
                bad.append((key, "zero-bias", np.abs(got).max() / wscale))
        else:
            l2 = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
            cos = (got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum())
            if not (l2 <= 0.15 and cos >= 0.99):
                bad.append((key, l2, cos))
        checked += 1
    assert not bad, bad
    assert checked == sum(1 for k in gold if k.startswith("grad."))
    # the un-pasted form of the terms has no kernel: loud, not another loss (advisor, round 4)
    T.opts.gen.p.paste_original_content = False
    with pytest.raises(NotImplementedError):
        T.get_painter_loss(batch)


def test_painter_local_global_discriminator_pair_matches_reference():
    """``dis.p.use_local_discriminator`` (reference trainer.py:1323-1358 on the G side, 1085-1099 on the D side; the pair is
    built by OmniDiscriminator, discriminator.py:246-252) against the reference's own modules (golden ``gstep_p_local``,
    lambdas.G.p.gan = 2 so that this branch's scaling shows): G-side terms and the gradient of every trainable Painter
    tensor; then, continuing from the spectral-norm state the G side left (as update_D follows update_G), the two D losses
    and the gradient of every trainable tensor of both discriminators."""
    name = "gstep_p_local"
    case = golden_cases()[name]
    gold = load_golden(name)
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer
    from helpers import disc_p_shapes
    from climategan_amd import fill

    opts = default_opts()
    opts.tasks = ["p"]
    opts.gen.p.latent_dim, opts.gen.p.spade_n_up = case["latent_dim"], case["n_up"]
    opts.dis.p.ndf, opts.dis.p.n_layers, opts.dis.p.num_D = case["ndf"], case["n_layers"], case["num_D"]
    opts.dis.p.use_local_discriminator = True
    opts.dis.soft_shift, opts.dis.flip_prob = 0.0, 0.0
    opts.train.lambdas.G.p.vgg = 0
    opts.train.lambdas.G.p.gan = case["local"]["lambda_gan"]
    T = Trainer(opts, device="cuda").setup(inference=False)
    assert set(T.D["p"].keys()) == {"global", "local"}
    T.G.painter.load_state_dict(case_state_dict(case), strict=True)
    shapes = disc_p_shapes(3, case["ndf"], case["n_layers"], case["num_D"])
    for i, which in enumerate(("global", "local")):
        T.D["p"][which].load_state_dict({k: t(v) for k, v in fill.fill_state_dict(shapes, case["seed"] + 1 + i).items()},
                                        strict=True)
    T.G.set_compute_dtype(torch.float16)
    T.D.set_compute_dtype(torch.float16)
    T.G.painter.set_latent_shape((case["B"], 3, case["H"], case["W"]), True)
    inp = {k: t(v).cuda() for k, v in case_inputs(name, case).items()}
    batch = {"rf": {"data": {"x": inp["x"], "m": inp["m"]}}}
    # ---- G side
    for p in T.D.parameters():
        p.requires_grad_(False)
    loss = T.get_painter_loss(batch)
    loss.backward()
    for key, log, tol in (("gan", "G.p.gan", 5e-3), ("featmatch", "G.p.featmatch", 1e-2)):
        ref, got = float(gold[key][0]), float(T.loss_log[log])
        assert abs(got - ref) <= tol * abs(ref), (key, got, ref)
    assert abs(loss.item() - float(gold["loss"][0])) <= 1e-2 * float(gold["loss"][0])

    def compare(named, prefix):
        bad, checked = [], 0
        for key, p in named:
            if not p.requires_grad or key.endswith(("weight_u", "weight_v")):
                continue
            assert p.grad is not None, key
            ref = gold[prefix + key].astype(np.float64)
            got = p.grad.cpu().numpy().astype(np.float64)
            base = key.rsplit(".", 1)[0]
            wkey = prefix + base + (".weight_bar" if prefix + base + ".weight_bar" in gold else ".weight")
            wscale = np.abs(gold[wkey]).max()
            if key.endswith("bias") and np.abs(ref).max() < 1e-4 * wscale:
                if np.abs(got).max() > 1e-2 * wscale:
                    bad.append((key, "zero-bias", np.abs(got).max() / wscale))
            else:
                l2 = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
                cos = (got * ref).sum() / np.sqrt((got ** 2).sum() * (ref ** 2).sum())
                if not (l2 <= 0.15 and cos >= 0.99):
                    bad.append((key, l2, cos))
            checked += 1
        assert not bad, (prefix, len(bad), [(k, "%.3g" % a if not isinstance(a, str) else a, "%.4f" % b) for k, a, b in bad])
        return checked

    assert compare(T.G.painter.named_parameters(), "grad.") == sum(1 for k in gold if k.startswith("grad."))
    # ---- pl4m with the pair (trainer.py:1628-1636): from the same discriminator state, the un-scaled sum of the two GAN terms
    # = the golden G-side GAN term / lambda; the mask (a prediction there) receives a gradient
    def load_d():
        for i, which in enumerate(("global", "local")):
            T.D["p"][which].load_state_dict({k: t(v) for k, v in fill.fill_state_dict(shapes, case["seed"] + 1 + i).items()},
                                            strict=True)
    post = {k: v.clone() for k, v in T.D.state_dict().items()}
    load_d()
    m_pred = inp["m"].clone().requires_grad_(True)
    pl4m = T.painter_loss_for_masker(inp["x"], m_pred)
    ref = float(gold["gan"][0]) / case["local"]["lambda_gan"]
    assert abs(pl4m.item() - ref) <= 1e-2 * ref, (pl4m.item(), ref)
    pl4m.backward()
    assert torch.isfinite(m_pred.grad).all() and m_pred.grad.abs().max() > 0
    assert all(p.requires_grad for k, p in T.G.painter.named_parameters() if not k.endswith(("weight_u", "weight_v")))
    T.D.load_state_dict(post)
    # ---- D side
    for key, p in T.D.named_parameters():
        if not key.endswith(("weight_u", "weight_v")):
            p.requires_grad_(True)
    # (on the reference's own painted image, as the golden D-step ``dstep_p`` does: a discriminator's first-layer gradients
    # are differences of nearly cancelling real / fake sums, and the 16-bit Painter's 0.5 % deviation of ``fake`` would be
    # what the comparison measures)
    gold_fake = t(gold["fake"]).cuda()
    T.G.paint = lambda m, x, **kw: gold_fake
    d_loss = T.get_D_loss(batch)
    del T.G.paint
    d_loss.backward()
    for which in ("global", "local"):
        ref, got = float(gold["d." + which][0]), float(T.loss_log["D.p." + which])
        assert abs(got - ref) <= 5e-3 * abs(ref), (which, got, ref)
        n = compare(T.D["p"][which].named_parameters(), "dgrad.%s." % which)
        assert n == sum(1 for k in gold if k.startswith("dgrad.%s." % which))
    # the other non-default switch of this range raises instead of being ignored
    T.opts.gen.p.diff_aug.use = True
    with pytest.raises(NotImplementedError):
        T.get_painter_loss(batch)


def test_painter_train_steps_run_and_learn():
    """Trainer.train_step (G update, D update, ExtraAdam extrapolate / step) incl. the VGG term with its random-init
    feature extractor: finite losses, parameters move, spectral-norm vectors advance, D flags restored."""
    case = golden_cases()[GNAME]
    T = build_trainer(case, torch.float16, vgg=True)
    inp = {k: t(v).cuda() for k, v in case_inputs(GNAME, case).items()}
    batch = {"rf": {"data": {"x": inp["x"], "m": inp["m"]}}}
    w0 = T.G.painter.conv_img.weight.detach().clone()
    u0 = T.G.painter.head_0.conv_0.module.weight_u.detach().clone()
    for _ in range(2):
        g, d = T.train_step(batch)
        assert torch.isfinite(g) and torch.isfinite(d)
    assert T.global_step == 2
    assert not torch.equal(T.G.painter.conv_img.weight.detach(), w0)
    assert not torch.equal(T.G.painter.head_0.conv_0.module.weight_u.detach(), u0)
    assert "G.p.vgg" in T.loss_log and torch.isfinite(T.loss_log["G.p.vgg"])
    flags = {n: p.requires_grad for n, p in T.D.named_parameters()}
    assert all(v == (not (n.endswith("weight_u") or n.endswith("weight_v"))) for n, v in flags.items())


# ------------------------------------------------------------------------------------------------ masker G step
MNAME = "mstep"


def build_masker_trainer(case, dt=torch.bfloat16, use_spade=False, detach=True):
    from climategan_amd import fill
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    opts = default_opts()
    opts.tasks = ["d", "s", "m"]
    if use_spade:
        opts.gen.m.use_spade = True
        opts.gen.m.spade.detach = detach
    T = Trainer(opts, device="cuda").setup(inference=False)
    for mod, seed in ((T.G, case["seed"]), (T.D, case["seed"] + 1)):
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed, gain=case["gain"]).items()})
    T.G.set_compute_dtype(dt)
    T.D.set_compute_dtype(dt)
    T.G.decoders["d"]._target_size = case["W"] // 4
    T.G.decoders["s"].set_target_size((case["H"] // 4, case["W"] // 4))
    return T


def masker_batch(case, name=MNAME):
    inp = {k: t(v).cuda() for k, v in case_inputs(name, case).items()}
    return {dom: {"data": {"x": inp["x_" + dom], "d": inp["d_" + dom], "s": inp["s_" + dom], "m": inp["m_" + dom]}}
            for dom in ("r", "s")}


# Cosine between the REFERENCE's fp32 gradients and the reference's own gradients when every conv / norm / activation
# output and the gradient flowing back through it is rounded to 16 bit (tests/devtools/measure_ref_grad_quant.py, dev container):
# this untrained ResNet-101 in training mode is chaotic enough that 16-bit storage alone decorrelates the encoder's
# gradient direction (norms stay within 4 %), for the reference exactly as for this build.
REF_QUANT_COS = {
    "float16": {"encoder.conv1.weight": 0.18, "encoder.layer3.16.conv3.weight": 0.30, "encoder.layer4.2.conv2.weight": 0.65,
                "decoders.m.model.4.conv.module.weight_bar": 0.9999},
    "bfloat16": {"encoder.conv1.weight": 0.00, "encoder.layer3.16.conv3.weight": 0.02, "encoder.layer4.2.conv2.weight": 0.36,
                 "decoders.d.enc4_2.conv.weight": 0.19, "decoders.s.aspp.conv1.conv.weight": 0.39,
                 "decoders.s.decoder.conv_cat.0.conv.weight": 0.84, "decoders.m.model.4.conv.module.weight_bar": 0.9997},
}


MASKER_TERMS = {"term.s.minent.r": "G.s.minent.r", "term.s.advent.r": "G.s.advent.r", "term.m.tv.r": "G.m.tv.r",
                "term.m.gi.r": "G.m.gi.r", "term.m.minent.r": "G.m.minent.r", "term.m.advent.r": "G.m.advent.r",
                "term.s.crossent.s": "G.s.crossent.s", "term.m.tv.s": "G.m.tv.s", "term.m.bce.s": "G.m.bce.s",
                "term.d.s": "G.d.s"}


def _check_masker_terms(T, gold, loss, rel, gi_abs):
    for gk, hk in MASKER_TERMS.items():
        ref, got = float(gold[gk][0]), float(T.loss_log[hk])
        if gk == "term.m.gi.r":                                   # GI counts pixels across a 0.5 threshold
            assert abs(got - ref) <= max(gi_abs, 0.25 * abs(ref)), (gk, got, ref)
        else:
            assert abs(got - ref) <= rel * max(abs(ref), 1e-4), (gk, got, ref)
    assert abs(loss.item() - float(gold["loss"][0])) <= rel * abs(float(gold["loss"][0]))


@pytest.mark.parametrize("name", [MNAME, "mstep_spade"])
def test_masker_loss_terms_fp16(name):
    """The ten loss terms of the golden masker steps (base and SPADE mask decoder) with fp16 storage (10-bit mantissa:
    the forward error of the chaotic untrained encoder stays small), within 2.5 % of the reference's fp32 values."""
    case = golden_cases()[name]
    gold = load_golden(name)
    T = build_masker_trainer(case, torch.float16, use_spade=bool(case.get("use_spade")), detach=False)
    with torch.no_grad():
        loss = T.get_masker_loss(masker_batch(case, name))
    _check_masker_terms(T, gold, loss, rel=2.5e-2, gi_abs=3e-5)


# The same measurement for the SPADE mask decoder (tests/devtools/measure_ref_grad_quant.py bf16|fp16 spade).
REF_QUANT_COS_SPADE = {
    "float16": {"decoders.m.low_level_conv.conv.module.weight_bar": 0.51, "decoders.m.spade_blocks.0.conv_0.module.weight_bar": 0.47,
                "decoders.m.spade_blocks.1.conv_0.module.weight_bar": 0.74, "decoders.m.spade_blocks.2.conv_0.module.weight_bar": 0.97,
                "decoders.m.spade_blocks.2.conv_1.module.weight_bar": 0.989, "decoders.m.mask_conv.conv.module.weight_bar": 0.998},
    "bfloat16": {"decoders.m.low_level_conv.conv.module.weight_bar": 0.08, "decoders.m.spade_blocks.0.conv_0.module.weight_bar": 0.11,
                 "decoders.m.spade_blocks.1.conv_0.module.weight_bar": 0.54, "decoders.m.spade_blocks.2.conv_0.module.weight_bar": 0.91,
                 "decoders.m.spade_blocks.2.conv_1.module.weight_bar": 0.981, "decoders.m.mask_conv.conv.module.weight_bar": 0.993},
}


@pytest.mark.parametrize("name", [MNAME, "mstep_spade"])
def test_masker_g_step_matches_reference(name):
    """get_masker_loss + backward on the HIP path (ResNet-101 with batch-statistics BatchNorm, DADA depth, DeepLab-v3+
    seg, mask decoder, frozen ADVENT discriminators, 10 loss terms over a real and a sim batch) vs the reference's own
    modules / loss classes / backward (golden ``mstep``), in bf16.

    What can be compared: every loss term; the NORM of every parameter gradient (16-bit storage does not move it: the
    reference's own 16-bit-rounded run keeps norms within 4 %); the gradient DIRECTION where 16-bit storage preserves
    it in the reference itself (mask decoder >= 0.99, end of the seg decoder >= 0.84, see REF_QUANT_COS): deeper in
    this untrained network the reference's own direction is lost too (encoder cosine 0.0 - 0.4), so no bound is put
    there.  Also the running statistics of four BatchNorm layers and the mask decoder's spectral-norm vectors.
    ``mstep_spade``: the same step with the SPADE mask decoder conditioned on the non-detached depth / segmentation
    predictions (defaults.yaml:168,182): batch-statistics SPADE, spectral_batch projections, make_m_cond's backward;
    its BatchNorm running statistics are compared too."""
    from climategan_amd import fill

    case = golden_cases()[name]
    gold = load_golden(name)
    T = build_masker_trainer(case, use_spade=bool(case.get("use_spade")), detach=False)
    for p in T.D.parameters():
        p.requires_grad_(False)
    loss = T.get_masker_loss(masker_batch(case, name))
    loss.backward()
    # bf16 through 33 training-mode bottlenecks of an untrained network: a few per cent on the loss terms, and which
    # way depends on every rounding on the way (fusing the residual add into the BatchNorm pass moved G.d.s from +1.8 %
    # to +5.5 % in bf16 and from +0.4 % to +0.2 % in fp16); test_masker_loss_terms_fp16 is the tight pin.
    _check_masker_terms(T, gold, loss, rel=8e-2, gi_abs=1e-4)
    params = dict(T.G.named_parameters())
    ratios, cos = {}, {}
    for gk in gold:
        if not gk.startswith("gsub."):
            continue
        key = gk[5:]
        g = params[key].grad
        assert g is not None and torch.isfinite(g).all(), key
        rn = float(gold["gnorm." + key][0])
        ref = gold[gk].astype(np.float64)
        if rn < 1e-7 or np.abs(ref).max() < 1e-9:
            continue                                              # exactly-zero gradients (biases in front of a norm)
        flat = g.reshape(-1).float().cpu().numpy()
        n = case["sub"]
        if flat.size > n:
            idx = (fill.uniform01((n,), fill.key_seed(key, 4242)) * flat.size).astype(np.int64).clip(0, flat.size - 1)
            sub = flat[idx].astype(np.float64)
        else:
            sub = flat.astype(np.float64)
        ratios[key] = float(np.linalg.norm(flat)) / rn
        cos[key] = (sub * ref).sum() / max(np.sqrt((sub ** 2).sum() * (ref ** 2).sum()), 1e-30)
    big = {k: v for k, v in ratios.items() if k.endswith("weight") or k.endswith("weight_bar")}
    assert len(big) > 200
    r = np.array(list(big.values()))
    # The encoder's gradient norms move together with the depth term that dominates the loss: 0.96 / 1.10 in bf16 and
    # 1.00 / 0.97 in fp16 for the unfused / fused bottleneck tail (same algebra, one rounding apart); decoders 1.00-1.01.
    assert 0.88 <= np.median(r) <= 1.15, np.median(r)
    dec = np.array([v for k, v in big.items() if not k.startswith("encoder.")])
    assert 0.95 <= np.median(dec) <= 1.06, np.median(dec)
    assert np.mean((r > 0.75) & (r < 1.35)) >= 0.95, sorted(big.items(), key=lambda kv: abs(np.log(kv[1])))[-8:]
    if case.get("use_spade"):
        # three training-mode batch-norm SPADE blocks behind chaotic conditioning maps: the reference's own 16-bit run
        # keeps the direction at the end of the decoder only (REF_QUANT_COS_SPADE); this build: 0.98-0.995 there,
        # 0.27-0.5 (bf16) / 0.5-0.65 (fp16) at block 0, above the reference's own 0.08-0.17 / 0.44-0.51
        for k, ref_cos in REF_QUANT_COS_SPADE["bfloat16"].items():
            if ref_cos >= 0.97:
                assert cos[k] >= 0.95, (k, cos[k])
        assert cos["decoders.m.spade_blocks.0.conv_0.module.weight_bar"] >= 0.15
    else:
        mdec = [v for k, v in cos.items() if k.startswith("decoders.m.") and k.endswith("weight_bar")]
        assert len(mdec) >= 10 and np.median(mdec) >= 0.95 and min(mdec) >= 0.85, (np.median(mdec), min(mdec))
    sdec = [v for k, v in cos.items() if k.startswith("decoders.s.decoder.conv_cat") and k.endswith("conv.weight")]
    assert min(sdec) >= 0.75, sdec
    sd = T.G.state_dict()
    for k in gold:
        if k.startswith("post."):
            ref = gold[k]
            # running statistics after two bf16 forwards (momentum 0.1): the deeper the layer the larger the forward error
            tol = 2e-5 if k.endswith("weight_u") else 1e-1 * max(np.abs(ref).max(), 1e-3)
            assert np.abs(sd[k[5:]].float().cpu().numpy() - ref).max() <= tol, k


def test_masker_train_step_runs():
    """update_G + update_D on a two-domain masker batch: finite losses, parameters and running statistics move."""
    case = golden_cases()[MNAME]
    T = build_masker_trainer(case)
    batch = masker_batch(case)
    w0 = T.G.encoder.layer4[2].conv3.weight.detach().clone()
    rm0 = T.G.encoder.bn1.running_mean.detach().clone()
    dw0 = T.D["s"]["Advent"][0].module.weight_bar.detach().clone()
    for _ in range(2):
        g, d = T.train_step(batch)
        assert torch.isfinite(g) and torch.isfinite(d)
    assert not torch.equal(T.G.encoder.layer4[2].conv3.weight.detach(), w0)
    assert not torch.equal(T.G.encoder.bn1.running_mean.detach(), rm0)
    assert not torch.equal(T.D["s"]["Advent"][0].module.weight_bar.detach(), dw0)


def test_forward_uses_the_parameters_the_optimizer_wrote():
    """The HIP optimizer and the BatchNorm kernels write parameters / running statistics through raw pointers; the
    packed-weight caches key on ``tensor._version``.  After train steps the trained modules' forward must equal, bit for
    bit, the forward of FRESH modules (empty caches) loaded from their state dict -- in training mode (packed conv
    weights) and in eval mode (BatchNorm folded from the running statistics)."""
    from climategan_amd.generator import create_generator

    case = golden_cases()[MNAME]
    T = build_masker_trainer(case)
    batch = masker_batch(case)
    x = batch["r"]["data"]["x"]
    with torch.no_grad():
        before = {k: v.clone() for k, v in T.G.masker_forward(x).items()}
    for _ in range(2):
        T.train_step(batch)
    sd = {k: v.detach().clone() for k, v in T.G.state_dict().items()}
    fresh = create_generator(T.opts, device="cuda", no_init=True)
    fresh.load_state_dict(sd)
    fresh.set_compute_dtype(torch.bfloat16)
    fresh.decoders["d"]._target_size = case["W"] // 4
    fresh.decoders["s"].set_target_size((case["H"] // 4, case["W"] // 4))
    for mode in ("train", "eval"):
        # no load_state_dict into T.G here: copy_ would bump the version counters and hide a stale cache.  Both
        # generators go through the same sequence of forwards from the same state, so their running statistics and
        # spectral-norm vectors stay identical.
        getattr(T.G, mode)()
        getattr(fresh, mode)()
        with torch.no_grad():
            outs = [G.masker_forward(x) for G in (T.G, fresh)]
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), (mode, k, (outs[0][k] - outs[1][k]).abs().max().item())
        if mode == "train":                # and the update is visible in the output at all
            assert all(not torch.equal(outs[0][k], before[k]) for k in before)


def test_masker_spade_decoder_train_step():
    """The SPADE mask decoder (gen.m.use_spade, batch-norm SPADE blocks conditioned on the DETACHED depth / seg / image
    map) trains: two update_G + update_D steps, finite losses, the decoder's parameters and its BatchNorm running
    statistics move, and the gradient reaches the encoder through z.  With the non-detached map (the reference's default)
    the mask terms also reach the depth / segmentation decoders."""
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    case = golden_cases()[MNAME]
    T = build_masker_trainer(case, use_spade=True)
    batch = masker_batch(case)
    dec = T.G.decoders["m"]
    w0 = dec.spade_blocks[0].norm_0.mlp_gamma.weight.detach().clone()
    p0 = dec.merge_feats_conv.conv.module.weight_bar.detach().clone()
    rm0 = dec.spade_blocks[1].norm_1.param_free_norm.running_mean.detach().clone()
    loss = T.get_masker_loss(batch)
    assert torch.isfinite(loss)
    T.G.zero_grad()
    loss.backward()
    g = dec.spade_blocks[0].norm_0.mlp_shared[0].weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    ge = T.G.encoder.layer4[2].conv3.weight.grad
    assert ge is not None and torch.isfinite(ge).all() and ge.abs().max() > 0
    T.G.zero_grad()
    for _ in range(2):
        gl, dl = T.train_step(batch)
        assert torch.isfinite(gl) and torch.isfinite(dl)
    assert not torch.equal(dec.spade_blocks[0].norm_0.mlp_gamma.weight.detach(), w0)
    assert not torch.equal(dec.merge_feats_conv.conv.module.weight_bar.detach(), p0)
    assert not torch.equal(dec.spade_blocks[1].norm_1.param_free_norm.running_mean.detach(), rm0)

    # the reference's default: the conditioning map is NOT detached (defaults.yaml:182) -- the mask losses reach the
    # depth and segmentation decoders through make_m_cond as well
    def decoder_grads(detach):
        T2 = build_masker_trainer(case, use_spade=True)
        T2.opts.gen.m.spade.detach = detach
        for k in ("d", "s"):
            for terms in (T2.opts.train.lambdas.G[k],):
                for name in list(terms.keys()):
                    terms[name] = 0                       # only the mask terms: d / s gradients can only come via cond
        T2.G.zero_grad()
        T2.get_masker_loss(batch).backward()
        gd = T2.G.decoders["d"].enc4_2.conv.weight.grad
        gs = T2.G.decoders["s"].decoder.conv_cat[0].conv.weight.grad
        # the mask terms alone must reach the projection convs and, through z, the encoder (the latent's bilinear
        # resize and concat are differentiable)
        for g in (T2.G.decoders["m"].low_level_conv.conv.module.weight_bar.grad, T2.G.encoder.layer1[0].conv1.weight.grad,
                  T2.G.encoder.layer4[2].conv3.weight.grad):
            assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
        return gd, gs

    gd0, gs0 = decoder_grads(True)
    gd1, gs1 = decoder_grads(False)
    for g0 in (gd0, gs0):
        assert g0 is None or g0.abs().max() == 0
    for g1 in (gd1, gs1):
        assert g1 is not None and torch.isfinite(g1).all() and g1.abs().max() > 0
    T3 = build_masker_trainer(case, use_spade=True)
    T3.opts.gen.m.spade.detach = False
    for _ in range(2):
        gl, dl = T3.train_step(batch)
        assert torch.isfinite(gl) and torch.isfinite(dl)


def test_gradient_reducer_over_rccl_single_rank():
    """The data-parallel path (broadcast of the replicas, bucketed all-reduce from post-accumulate-grad hooks, finish()
    before extrapolation / step) on a ONE-rank RCCL group: the only way to run the RCCL code on a single-GPU box.  With
    one rank the average is the identity, so the step must reproduce the reducer-free step (up to the run-to-run jitter
    of the fp32 atomics in the bias-gradient and loss reductions)."""
    import os
    import torch.distributed as dist

    case = golden_cases()[GNAME]
    inp = {k: t(v).cuda() for k, v in case_inputs(GNAME, case).items()}
    batch = {"rf": {"data": {"x": inp["x"], "m": inp["m"]}}}

    def run():
        T = build_trainer(case, torch.bfloat16)
        outs = [T.train_step(batch) for _ in range(2)]
        return T, outs

    T0, ref = run()
    assert T0.g_reducer is None
    import socket
    with socket.socket() as sk:                       # a free port: nothing else may be listening on a fixed one
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), CGAN_DDP_SINGLE_RANK_TEST="1")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        T1, got = run()
        assert T1.g_reducer is not None and T1.g_reducer.active and len(T1.g_reducer.buckets) >= 1
        assert all(b.work is None and b.pending == len(b.params) for b in T1.g_reducer.buckets)   # re-armed by finish()
        for (g0, d0), (g1, d1) in zip(ref, got):
            assert torch.allclose(g0, g1, rtol=2e-3, atol=1e-5) and torch.allclose(d0, d1, rtol=2e-3, atol=1e-5)
        # fp16-style loss scaling under the reducer: the buckets carry the still-scaled gradients (they are
        # launched from hooks during the backward); finish() writes the averages back and only THEN the scale is divided
        # out -- the gradients the optimizer sees must be those of the unscaled run (bf16 wire format: 2^-7 relative)
        from climategan_amd import autograd as ag
        assert T1.d_reducer.grad_dtype == torch.float32            # exact exchange by default; bf16 buckets are opt-in

        def d_grads(scale):
            ag.set_grad_scale(scale)
            os.environ["CGAN_DDP_BF16_GRADS"] = "1"                # ... and exercised here
            try:
                T = build_trainer(case, torch.bfloat16)
                assert T.d_reducer.grad_dtype == torch.bfloat16
                T.update_D(batch)
                return {k: p.grad.detach().clone() for k, p in T.D.named_parameters() if p.grad is not None}
            finally:
                ag.set_grad_scale(1.0)
                os.environ.pop("CGAN_DDP_BF16_GRADS", None)

        g1, g256 = d_grads(1.0), d_grads(256.0)
        assert set(g1) == set(g256) and len(g1) > 10
        for k in g1:
            scale = g1[k].abs().max().item()
            assert (g1[k] - g256[k]).abs().max().item() <= 2.0 ** -6 * scale + 1e-12, k
        # the Masker trainer (17 G buckets at the default sizes, 3 discriminators) under the same group: every bucket
        # of G and D must fill from the hooks or be completed by finish() -- a parameter that got no gradient while
        # its bucket-mates did would raise
        mcase = golden_cases()[MNAME]
        TM = build_masker_trainer(mcase)
        assert TM.g_reducer.active and len(TM.g_reducer.buckets) > 4
        for _ in range(2):
            g, d = TM.train_step(masker_batch(mcase))
            assert torch.isfinite(g) and torch.isfinite(d)
        # the same exchange WITHOUT torch in the collective (CGAN_DDP_DIRECT_RCCL=1): the reducer's own communicator
        # through the C ABI (cgan_rccl_load / cgan_comm_unique_id / cgan_comm_init_rank), cgan_allreduce_bucket on its own
        # stream between two events; with one rank the sum is the identity, so the steps must match the reducer-free ones
        os.environ["CGAN_DDP_DIRECT_RCCL"] = "1"
        try:
            T2, got2 = run()
            assert T2.g_reducer.direct and T2.d_reducer.direct and T2.g_reducer._comm is not None
            for (g0, d0), (g1, d1) in zip(ref, got2):
                assert torch.allclose(g0, g1, rtol=2e-3, atol=1e-5) and torch.allclose(d0, d1, rtol=2e-3, atol=1e-5)
            for (k, a), (_, b) in zip(T0.G.state_dict().items(), T2.G.state_dict().items()):
                assert torch.allclose(a.float(), b.float(), rtol=1e-3, atol=3e-4), k
            # the entry point on a buffer of its own: sum over one rank = the buffer, in fp32 and on the bf16 wire
            from climategan_amd import _lib
            lib = _lib.load()
            for dt, code in ((torch.float32, _lib.CGAN_F32), (torch.bfloat16, _lib.CGAN_BF16)):
                buf = torch.randn(1 << 20, device="cuda").to(dt)
                keep = buf.clone()
                _lib.check(lib.cgan_allreduce_bucket(buf.data_ptr(), buf.numel(), code, T2.g_reducer._comm,
                                                     torch.cuda.current_stream().cuda_stream), "cgan_allreduce_bucket")
                torch.cuda.synchronize()
                assert torch.equal(buf, keep)
            T2.g_reducer.remove()
            T2.d_reducer.remove()
            assert T2.g_reducer._comm is None
        finally:
            os.environ.pop("CGAN_DDP_DIRECT_RCCL", None)
        for mod0, mod1 in ((T0.G, T1.G), (T0.D, T1.D)):
            for (k, a), (_, b) in zip(mod0.state_dict().items(), mod1.state_dict().items()):
                # Adam moves every weight by ~lr per update whatever the gradient's scale: tensors whose true gradient is
                # zero (biases in front of an instance norm) follow the sign of rounding noise, so two runs may differ
                # by a few lr (5e-5) there
                assert torch.allclose(a.float(), b.float(), rtol=1e-3, atol=3e-4), k
    finally:
        dist.destroy_process_group()
        os.environ.pop("CGAN_DDP_SINGLE_RANK_TEST", None)


def test_fp16_loss_scale_is_divided_out():
    """autograd.set_grad_scale(S): the activation gradients travel S times larger (fp16 range), the parameter gradients
    the optimizer sees do not (Adam would hide a forgotten unscale: it is invariant to the gradient's scale)."""
    from climategan_amd import autograd as ag

    case = golden_cases()[GNAME]
    inp = {k: t(v).cuda() for k, v in case_inputs(GNAME, case).items()}
    batch = {"rf": {"data": {"x": inp["x"], "m": inp["m"]}}}

    def d_grads(scale):
        ag.set_grad_scale(scale)
        try:
            T = build_trainer(case, torch.float16)
            seen = {}
            T.d_opt.extrapolation = lambda: seen.update({n: p.grad.detach().clone() for n, p in T.D.named_parameters()
                                                          if p.grad is not None})
            T.update_D(batch)
            return seen
        finally:
            ag.set_grad_scale(1.0)

    g1, g64 = d_grads(1.0), d_grads(64.0)
    assert g1.keys() == g64.keys() and len(g1) > 10
    gmax = max(v.abs().max().item() for v in g1.values())
    for k in g1:
        a, b = g1[k].float(), g64[k].float()
        # same magnitude (not 64x), up to 16-bit rounding; biases in front of an instance norm hold pure rounding noise
        assert (a - b).abs().max().item() <= 0.1 * a.abs().max().item() + 3e-4 * gmax, k


def test_run_evaluation_and_eval_images():
    """``Trainer.run_evaluation`` / ``eval_images`` (reference trainer.py:1653-1799): eval mode and no_grad around
    ``get_G_loss`` per validation batch with the logged terms averaged, then accuracy / mIOU of the Masker's predictions over
    the display images -- one image at a time, the table the reference prints; nothing trains, the trainer is back in train
    mode afterwards.  The metric values are checked against the same metrics computed here from ``G.masker_forward`` (the
    metric kernel itself is pinned by tests/test_gpu_metrics.py against the reference's eval_metrics.py)."""
    from climategan_amd import eval_metrics

    case = golden_cases()[MNAME]
    T = build_masker_trainer(case, dt=torch.float16)
    batch = masker_batch(case)
    n = batch["r"]["data"]["x"].shape[0]
    hs, ws = case["H"] // 4, case["W"] // 4
    images = {"val": {dom: [{"data": {"x": batch[dom]["data"]["x"][i].float(),
                                       "s": batch[dom]["data"]["s"][i].reshape(1, hs, ws),
                                       "m": (batch[dom]["data"]["m"][i].reshape(1, case["H"], case["W"]) > 0.5).long()}}
                             for i in range(n)] for dom in ("r", "s")}}
    before = {k: v.clone() for k, v in T.G.state_dict().items()}
    half = {dom: {"data": {k: v[: max(n // 2, 1)] for k, v in batch[dom]["data"].items()}} for dom in batch}
    out = T.run_evaluation([batch, half], display_images=images)
    assert T.current_mode == "train" and T.G.training
    # validation does not train: parameters untouched (eval-mode BatchNorm: running statistics too), no gradients left behind
    after = T.G.state_dict()
    moved = [k for k in before if not torch.equal(before[k], after[k]) and not k.endswith(("weight_u", "weight_v"))]
    assert not moved, moved[:5]
    assert all(p.grad is None for p in T.G.parameters())
    # averaged generator terms: the mean over the two batches of what get_G_loss logs for each
    T.eval_mode()
    terms = []
    with torch.no_grad():
        for b in (batch, half):
            T.loss_log = {}
            assert torch.isfinite(T.get_G_loss(b))
            terms.append({k: float(v) for k, v in T.loss_log.items() if k.startswith("G.")})
    assert set(out["losses"]) == set(terms[0]) and len(terms[0]) >= 8
    for k, v in out["losses"].items():
        ref = 0.5 * (terms[0][k] + terms[1][k])
        # (the spectral-norm power iterations advance between the passes; GroundIntersection counts pixels across a threshold)
        assert abs(v - ref) <= (0.25 if ".gi." in k else 1e-2) * abs(ref) + 2e-5, (k, v, ref)
    # the metric table: tasks m and s, accuracy and mIOU, per domain
    assert set(out["metrics"]) == {"r", "s"}
    for dom in ("r", "s"):
        tab = out["metrics"][dom]
        assert set(tab) == {"m", "s"} and all(set(v) == {"accuracy", "mIOU"} for v in tab.values())
        acc_s, iou_s, acc_m, iou_m = [], [], [], []
        with torch.no_grad():
            for im in images["val"][dom]:
                x = im["data"]["x"].unsqueeze(0).cuda()
                pred = T.G.masker_forward(x)
                s, m = im["data"]["s"].unsqueeze(0).cuda(), im["data"]["m"].unsqueeze(0).cuda()
                acc_s.append(eval_metrics.accuracy(pred["s"].float(), s))
                iou_s.append(eval_metrics.mIOU(pred["s"].float(), s))
                pm = (pred["m"] > 0.5).float()
                acc_m.append(eval_metrics.accuracy(pm, m))
                iou_m.append(eval_metrics.mIOU(torch.cat([1 - pm, pm], 1), m))
        def mean(v):
            v = float(np.mean(v))
            return -1 if np.isnan(v) else v
        # (the mask decoder's spectral-norm vectors advance with every forward: a handful of pixels at the 0.5 threshold may flip)
        assert abs(tab["s"]["accuracy"] - mean(acc_s)) <= 2e-3 and abs(tab["s"]["mIOU"] - mean(iou_s)) <= 5e-3
        assert abs(tab["m"]["accuracy"] - mean(acc_m)) <= 2e-3 and abs(tab["m"]["mIOU"] - mean(iou_m)) <= 2e-2
        assert 0.0 <= tab["s"]["accuracy"] <= 1.0
    T.train_mode()
    # domains without display images, and the rf domain, are skipped like in the reference
    assert T.eval_images("val", "rf") is None and T.eval_images("train", "r") is None
