"""GPU parity of the masker-side losses (HIP value + gradient kernels behind climategan_amd.losses) against values and
input gradients computed by the REFERENCE's own loss classes (oracle/make_golden.py: masker_losses), composed the way
masker_s_loss / masker_m_loss compose them (softmax / sigmoid in front, trainer.py:1409-1616).

Inputs are fp32 in the reference and rounded to fp16 NHWC here: loss values within 2e-3 relative, gradients within
4e-3 of their scale (two to three 16-bit roundings: the probability map and the gradient flowing back through it)."""
import numpy as np
import pytest
import torch

from helpers import golden_cases, load_golden, t
from oracle.make_golden import case_inputs

pytestmark = pytest.mark.gpu
NAME = "masker_losses"


def nhwc(a, requires_grad=False):
    from climategan_amd import ops
    x = ops.nchw_to_nhwc(t(a).cuda(), torch.float16)
    if requires_grad:
        x.t.requires_grad_(True)
    return x


def grad_nchw(x):
    from climategan_amd import ops
    return ops.nhwc_to_nchw(ops.NHWC(x.t.grad, x.c)).cpu().numpy()


def check(name, gold, loss, x, vtol=2e-3, gtol=4e-3):
    ref = float(gold[name][0])
    assert abs(loss.item() - ref) <= vtol * max(abs(ref), 1e-3), (name, loss.item(), ref)
    if x is not None:
        g, gr = grad_nchw(x), gold[name + ".grad"]
        scale = max(np.abs(gr).max(), 1e-12)
        err = np.abs(g - gr).max()
        assert err <= gtol * scale, "%s grad: max err %.3g (scale %.3g)" % (name, err, scale)


def test_masker_losses_match_reference():
    from climategan_amd import losses as L

    case = golden_cases()[NAME]
    gold = load_golden(NAME)
    inp = case_inputs(NAME, case)

    s = nhwc(inp["s_logits"], True)
    loss = L.CrossEntropy()(s, t(inp["s_target"]).cuda())
    loss.backward()
    check("crossent", gold, loss, s)

    s = nhwc(inp["s_logits"], True)
    loss = L.MinentLoss()(L.softmax(s))
    loss.backward()
    check("minent_v1", gold, loss, s)

    s = nhwc(inp["s_logits"], True)
    ent = L.prob_2_entropy(L.softmax(s), nhwc(inp["d_pred"]))
    from climategan_amd import ops
    got_ent = ops.nhwc_to_nchw(ops.NHWC(ent.t.detach(), ent.c)).cpu().numpy()
    assert np.abs(got_ent - gold["entropy_dada"]).max() <= 2e-3 * np.abs(gold["entropy_dada"]).max()
    ent.t.backward(torch.full_like(ent.t, 0.37))       # d/ds of (ent * 0.37).sum()
    g, gr = grad_nchw(s), gold["entropy_dada_sum.grad"]
    assert np.abs(g - gr).max() <= 4e-3 * np.abs(gr).max()

    m = nhwc(inp["m_logits"], True)
    loss = L.BCEWithLogitsLoss()(m, t(inp["m_target"]).cuda())
    loss.backward()
    check("bce", gold, loss, m)

    m = nhwc(inp["m_logits"], True)
    prob = L.sigmoid_pair(m)                            # [p, 1 - p]
    loss = L.MinentLoss(version=2, lambda_var=0.1)(prob)
    loss.backward()
    check("minent_v2", gold, loss, m)

    m = nhwc(inp["m_logits"], True)
    p1 = L.sigmoid(m)
    loss = L.TVLoss()(p1)
    loss.backward()
    check("tv", gold, loss, m, gtol=8e-3)          # second differences of a 16-bit map

    loss = L.GroundIntersectionLoss()(p1, t(inp["ground"]).cuda())
    assert abs(loss.item() - float(gold["gi"][0])) <= 2e-3           # a few pixels sit within fp16 rounding of 0.5

    for y in (0, 1):
        d = nhwc(inp["d_out"], True)
        from climategan_amd.autograd import advent_wgan
        loss = advent_wgan(d, float(y))
        loss.backward()
        check("advent_wgan_%d" % y, gold, loss, d)


def test_tv_loss_gradient_matches_torch():
    """TVLoss value + gradient on a multi-channel map against torch autograd of the reference formula."""
    from climategan_amd import fill, losses as L, ops
    x = t(fill.uniform((2, 3, 17, 23), 9100)).half().float().requires_grad_(True)
    h_tv = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum()
    ref = 2 * (h_tv / (3 * 16 * 23) + w_tv / (3 * 17 * 22)) / 2
    ref.backward()
    xg = ops.nchw_to_nhwc(x.detach().cuda(), torch.float16)
    xg.t.requires_grad_(True)
    loss = L.TVLoss()(xg)
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-3 * abs(ref.item())
    g = ops.nhwc_to_nchw(ops.NHWC(xg.t.grad, 3)).cpu()
    assert (g - x.grad).abs().max() <= 2e-3 * x.grad.abs().max()


def test_sigm_loss_matches_reference():
    """SIGMLoss (medians by radix select, mean absolute deviations, 4-scale Sobel term with the reference's B-fold count)
    value and gradient vs the reference class.  The gradient has one special entry -- the median element, which receives
    the chain-rule terms through the median -- compared separately because ties (fp16 values) let torch pick another
    index of the same value."""
    from climategan_amd import losses as L
    case = golden_cases()[NAME]
    gold = load_golden(NAME)
    inp = case_inputs(NAME, case)
    x = nhwc(inp["depth_pred"], True)
    loss = L.SIGMLoss(0.5)(x, t(inp["depth_target"]).cuda())
    loss.backward()
    ref = float(gold["sigm"][0])
    assert abs(loss.item() - ref) <= 1e-3 * abs(ref), (loss.item(), ref)
    g, gr = grad_nchw(x).reshape(-1), gold["sigm.grad"].reshape(-1).copy()
    pred16 = t(inp["depth_pred"]).half().float().reshape(-1).numpy()
    med_ref = int(gold["sigm.median_index"][0])
    ties = np.nonzero(pred16 == pred16[med_ref])[0]                   # the median value's tie set
    mask = np.ones_like(g, dtype=bool)
    mask[ties] = False
    scale = np.abs(gr[mask]).max()
    assert np.abs(g[mask] - gr[mask]).max() <= 4e-3 * scale
    # the special entry: same total over the tie set (the extra median term lands on one of its members)
    assert abs(g[ties].sum() - gr[ties].sum()) <= 2e-2 * max(abs(gr[ties].sum()), scale)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_hinge_loss_matches_reference(dt):
    """A16 ``HingeLoss`` (gen.p.loss == "hinge"): D-real / D-fake / G values and per-scale gradients vs the reference's own
    class (golden ``hinge_small``).  The logits are multiples of 1/8, exact in both 16-bit types, so the only rounding is
    the gradient store (constants +-1/N or +-0.5/N on the exact ties at +-1): values to 1e-6, gradients to 2^-9."""
    from climategan_amd import losses as L
    from climategan_amd import ops

    case, gold = golden_cases()["hinge_small"], load_golden("hinge_small")
    inp = case_inputs("hinge_small", case)
    crit = L.HingeLoss()
    for tag, real, for_d in (("d_real", True, True), ("d_fake", False, True), ("g", True, False)):
        preds = []
        for i in range(len(case["sizes"])):
            x = ops.nchw_to_nhwc(t(inp["p%d" % i]).cuda(), dt)
            x.t.requires_grad_(True)
            preds.append(x)
        loss = crit([[None, p] for p in preds], real, for_d)
        loss.backward()
        ref = float(gold[tag][0])
        assert abs(loss.item() - ref) <= 1e-6 * max(1.0, abs(ref)), (tag, loss.item(), ref)
        for i, p in enumerate(preds):
            g = ops.nhwc_to_nchw(ops.NHWC(p.t.grad, 1)).cpu().numpy()
            gr = gold["%s.grad%d" % (tag, i)]
            assert np.abs(g - gr).max() <= 2.0 ** -8 * np.abs(gr).max(), (tag, i)
            assert (p.t.grad[..., 1:] == 0).all()
    with pytest.raises(AssertionError):
        crit(preds[0], False, False)


def test_get_losses_selects_hinge():
    from climategan_amd import losses as L
    from climategan_amd.config import default_opts

    o = default_opts()
    o.tasks = ["p"]
    o.gen.p.loss = "hinge"
    ls = L.get_losses(o, 0, "cuda")
    assert isinstance(ls["G"]["p"]["gan"], L.HingeLoss) and ls["D"]["p"] is ls["G"]["p"]["gan"]
    o.gen.p.loss = "gan"
    assert isinstance(L.get_losses(o, 0, "cuda")["G"]["p"]["gan"], L.GANLoss)
