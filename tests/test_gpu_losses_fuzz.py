"""Seeded random-shape sweep of the masker losses (HIP value + gradient kernels) against the oracle restatements
(oracle.cpu_ref, pinned on the reference's own loss classes by tests/test_oracle_golden.py): odd map sizes, one to three
samples, class counts other than 11, maps too small for some Sobel scales."""
import numpy as np
import pytest
import torch

from climategan_amd import fill
from oracle import cpu_ref
from test_gpu_losses import grad_nchw, nhwc

pytestmark = pytest.mark.gpu


def draw(n, seed):
    rng = np.random.RandomState(seed)
    return [(int(rng.choice([1, 2, 3])), int(rng.choice([3, 5, 11, 19])), int(rng.choice([8, 17, 24, 33, 40])),
             int(rng.choice([8, 12, 25, 32, 47]))) for _ in range(n)]


def compare(loss, ref_loss, x, ref_leaf, vtol=2e-3, gtol=6e-3):
    loss.backward()
    (gr,) = torch.autograd.grad(ref_loss, ref_leaf)
    assert abs(loss.item() - ref_loss.item()) <= vtol * max(abs(ref_loss.item()), 1e-3), (loss.item(), ref_loss.item())
    g, gr = grad_nchw(x), gr.numpy()
    assert np.abs(g - gr).max() <= gtol * max(np.abs(gr).max(), 1e-12)


@pytest.mark.parametrize("shape", draw(10, 31))
def test_segmentation_losses(shape):
    from climategan_amd import losses as L
    b, c, h, w = shape
    logits = fill.uniform((b, c, h, w), 7 * h + w, -3, 3)
    target = (fill.uniform01((b, h, w), 9 * h + w) * c).astype(np.int64).clip(0, c - 1)
    s16 = torch.from_numpy(logits).half().float()

    s = nhwc(logits, True)
    leaf = s16.clone().requires_grad_(True)
    compare(L.CrossEntropy()(s, torch.from_numpy(target).cuda()), cpu_ref.cross_entropy(leaf, torch.from_numpy(target)), s, leaf)

    s = nhwc(logits, True)
    leaf = s16.clone().requires_grad_(True)
    compare(L.MinentLoss()(L.softmax(s)), cpu_ref.minent_loss(torch.softmax(leaf, dim=1)), s, leaf)


@pytest.mark.parametrize("shape", draw(10, 32))
def test_mask_losses(shape):
    from climategan_amd import losses as L
    from climategan_amd.autograd import advent_wgan
    b, _, h, w = shape
    logits = fill.uniform((b, 1, h, w), 5 * h + w, -4, 4)
    m16 = torch.from_numpy(logits).half().float()

    m = nhwc(logits, True)
    leaf = m16.clone().requires_grad_(True)
    compare(L.TVLoss()(L.sigmoid(m)), cpu_ref.tv_loss(torch.sigmoid(leaf)), m, leaf, gtol=1.2e-2)

    m = nhwc(logits, True)
    leaf = m16.clone().requires_grad_(True)
    p = torch.sigmoid(leaf)
    compare(L.MinentLoss(version=2, lambda_var=0.1)(L.sigmoid_pair(m)), cpu_ref.minent_loss(torch.cat([p, 1 - p], 1), 2, 0.1),
            m, leaf)

    target = torch.from_numpy((fill.uniform01((b, 1, h, w), 3 * h + w) > 0.5).astype(np.float32))
    m = nhwc(logits, True)
    leaf = m16.clone().requires_grad_(True)
    compare(L.BCEWithLogitsLoss()(m, target.cuda()), torch.nn.functional.binary_cross_entropy_with_logits(leaf, target), m, leaf)

    for y in (0, 1):
        d = nhwc(logits, True)
        leaf = m16.clone().requires_grad_(True)
        compare(advent_wgan(d, float(y)), cpu_ref.advent_wgan(leaf, y), d, leaf)

    ground = torch.from_numpy((fill.uniform01((b, 1, h, w), 4 * h + w) > 0.6).astype(np.float32))
    gi = L.GroundIntersectionLoss()(L.sigmoid(nhwc(logits)), ground.cuda())
    ref = cpu_ref.ground_intersection_loss(torch.sigmoid(m16), ground)
    assert abs(gi.item() - ref.item()) <= 3.0 / (b * h * w) + 2e-3        # pixels within fp16 rounding of the 0.5 step


@pytest.mark.parametrize("shape", [s for s in draw(14, 33) if s[2] >= 24 and s[3] >= 24][:6])
def test_sigm_loss(shape):
    """Value and gradient away from the median's tie set (see tests/test_gpu_losses.py for why that entry is special)."""
    from climategan_amd import losses as L
    b, _, h, w = shape
    pred = fill.uniform((b, 1, h, w), 11 * h + w, 0.3, 7.0)
    targ = torch.from_numpy(fill.uniform((b, 1, h, w), 12 * h + w, 0.3, 7.0))
    p16 = torch.from_numpy(pred).half().float()
    x = nhwc(pred, True)
    loss = L.SIGMLoss(0.5)(x, targ.cuda())
    loss.backward()
    leaf = p16.clone().requires_grad_(True)
    ref = cpu_ref.sigm_loss(leaf, targ)
    (gr,) = torch.autograd.grad(ref, leaf)
    assert abs(loss.item() - ref.item()) <= 2e-3 * abs(ref.item()), (loss.item(), ref.item())
    g, gr = grad_nchw(x).reshape(-1), gr.numpy().reshape(-1)
    med = torch.median(p16.flatten()).item()
    keep = p16.flatten().numpy() != med
    assert np.abs(g[keep] - gr[keep]).max() <= 6e-3 * np.abs(gr[keep]).max()
