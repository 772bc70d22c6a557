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


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_advent_input_pair_from_logits(dt):
    """losses.advent_input: the ADVENT discriminators' input from the logits as a (hi | lo) pair.  hi + lo must equal
    prob_2_entropy(softmax(s)) * depth (and the mask form) of the SAME 16-bit logits to ~2^-16 (bf16) / 2^-21 (fp16) of the
    value -- one 16-bit entropy map would be off by 2^-9 / 2^-12 -- and agree with the reference's tensor (fp32 logits)
    within the logits' own rounding; the backward reads d(pair)'s hi half."""
    from climategan_amd import losses as L, ops
    from oracle import cpu_ref

    case = golden_cases()[NAME]
    gold = load_golden(NAME)
    inp = case_inputs(NAME, case)
    s = ops.nchw_to_nhwc(t(inp["s_logits"]).cuda(), dt)
    s.t.requires_grad_(True)
    d = ops.nchw_to_nhwc(t(inp["d_pred"]).cuda(), dt)
    pair = L.advent_input(s, d)
    C = s.c
    assert pair.c == 2 * C
    got = (pair.t[..., :C].float() + pair.t[..., C:2 * C].float()).permute(0, 3, 1, 2).cpu()
    assert (pair.t[..., 2 * C:] == 0).all()
    s16 = ops.nhwc_to_nchw(ops.NHWC(s.t.detach(), C)).float().cpu().requires_grad_(True)
    d16 = ops.nhwc_to_nchw(ops.NHWC(d.t, 1)).float().cpu()
    ref = cpu_ref.prob_2_entropy(torch.softmax(s16, 1)) * d16
    lo_ulp = 2.0 ** -16 if dt == torch.bfloat16 else 2.0 ** -21
    err = (got - ref.detach()).abs().max().item()
    single = (pair.t[..., :C].float().permute(0, 3, 1, 2).cpu() - ref.detach()).abs().max().item()
    print("\nadvent pair %s: |hi + lo - ref| max %.3g (hi alone %.3g), scale %.3g" % (dt, err, single, ref.abs().max().item()))
    assert err <= 2 * lo_ulp * ref.abs().max().item() + 2e-6          # + the fast exp / log2 of the kernel
    assert np.abs(got.detach().numpy() - gold["entropy_dada"]).max() <= (3e-2 if dt == torch.bfloat16 else 4e-3) * np.abs(gold["entropy_dada"]).max()
    # backward: upstream gradient on both halves (the data gradient of a conv with duplicated weights); only hi is read
    up = torch.zeros_like(pair.t)
    g = torch.randn(pair.t.shape[:-1] + (C,), device="cuda").to(dt)
    up[..., :C] = g
    up[..., C:2 * C] = g
    pair.t.backward(up)
    (ref * g.float().permute(0, 3, 1, 2).cpu()).sum().backward()
    mine = ops.nhwc_to_nchw(ops.NHWC(s.t.grad, C)).float().cpu()
    tol = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    assert (mine - s16.grad).abs().max().item() <= 2 * tol * s16.grad.abs().max().item()

    # the mask form: cat[sigmoid(x), 1 - sigmoid(x)] -> entropy, C = 2
    m = ops.nchw_to_nhwc(t(inp["m_logits"]).cuda(), dt)
    m.t.requires_grad_(True)
    pair = L.advent_input(m, None, sigmoid_pair=True)
    assert pair.c == 4
    got = (pair.t[..., :2].float() + pair.t[..., 2:4].float()).permute(0, 3, 1, 2).cpu()
    m16 = ops.nhwc_to_nchw(ops.NHWC(m.t.detach(), 1)).float().cpu().requires_grad_(True)
    p = torch.sigmoid(m16)
    ref = cpu_ref.prob_2_entropy(torch.cat([p, 1 - p], 1))
    assert (got - ref.detach()).abs().max().item() <= 2 * lo_ulp * ref.abs().max().item() + 2e-6
    g = torch.randn(pair.t.shape[:-1] + (2,), device="cuda").to(dt)
    up = torch.zeros_like(pair.t)
    up[..., :2] = g
    up[..., 2:4] = g
    pair.t.backward(up)
    (ref * g.float().permute(0, 3, 1, 2).cpu()).sum().backward()
    mine = ops.nhwc_to_nchw(ops.NHWC(m.t.grad, 1)).float().cpu()
    assert (mine - m16.grad).abs().max().item() <= 2 * tol * m16.grad.abs().max().item()


def test_fc_discriminator_on_pair_matches_single_map():
    """FCDiscriminator on the (hi | lo) pair = the same network on hi + lo: for an input whose lo half is zero the two
    forwards and all gradients agree bit for bit up to the wgrad summation order; spectral-norm state advances once."""
    from climategan_amd import ops
    from climategan_amd.discriminator import get_fc_discriminator

    torch.manual_seed(0)
    dt = torch.float16
    for use_norm in (True, False):
        D = get_fc_discriminator(num_classes=3, use_norm=use_norm).cuda()
        D.compute_dtype = dt
        D2 = get_fc_discriminator(num_classes=3, use_norm=use_norm).cuda()
        D2.compute_dtype = dt
        D2.load_state_dict(D.state_dict())
        x = torch.rand(2, 3, 64, 64, device="cuda")
        single = ops.nchw_to_nhwc(x, dt)
        pair_t = torch.zeros(2, 64, 64, 8, device="cuda", dtype=dt)
        pair_t[..., :3] = single.t[..., :3]
        single.t.requires_grad_(True)
        pair_t.requires_grad_(True)
        y1 = D(single, nhwc=True)
        y2 = D2(ops.NHWC(pair_t, 6), nhwc=True)
        assert torch.equal(y1.t, y2.t)
        y1.t.float().sum().backward()
        y2.t.float().sum().backward()
        # (round 5: the 3-channel input's data gradient runs as the sub-pixel 3x3 conv, the 6-channel pair's by parity classes:
        # the same products in another fp32 summation order -- within one rounding step of the 16-bit gradient)
        ga, gb = single.t.grad[..., :3].float(), pair_t.grad[..., :3].float()
        assert (ga - gb).abs().max().item() <= 2 ** -9 * gb.abs().max().item()
        assert torch.equal(pair_t.grad[..., :3], pair_t.grad[..., 3:6])
        for (k, p1), (_, p2) in zip(D.named_parameters(), D2.named_parameters()):
            assert torch.equal(p1.data, p2.data), k                       # u, v advanced identically
            if p1.grad is not None:
                scale = p1.grad.abs().max().item() + 1e-12
                assert (p1.grad - p2.grad).abs().max().item() <= 1e-5 * scale, k
