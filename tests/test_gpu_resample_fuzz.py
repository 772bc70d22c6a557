"""Seeded random-shape sweep of the re-sampling kernels (forward and backward) against torch: bilinear (both corner
conventions, up and down, non-integer ratios), bicubic, legacy nearest, the 3x3 / 2x2 pools, nearest x2 upsample.
Forward values are 16-bit outputs of fp32 arithmetic (1e-3 / 8e-3 of scale); max-pool routing is exact (untied
inputs)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from climategan_amd import fill

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}


def draw(n, seed):
    rng = np.random.RandomState(seed)
    return [(int(rng.choice([1, 2])), int(rng.choice([1, 3, 11, 24, 64])), int(rng.choice([2, 5, 8, 13, 20, 33])),
             int(rng.choice([2, 4, 9, 16, 27, 40])), int(rng.choice([1, 3, 7, 16, 25, 41, 64])),
             int(rng.choice([2, 5, 8, 19, 32, 50]))) for _ in range(n)]


def q(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).float()


def rel(got, ref):
    return (got - ref).abs().max().item() / max(ref.abs().max().item(), 1e-12)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", draw(12, 51))
def test_bilinear_forward_backward(dt, shape):
    from climategan_amd import ops
    from climategan_amd.autograd import ResizeBilinearFn
    b, c, h, w, oh, ow = shape
    for align in (False, True):
        x = q(fill.uniform((b, c, h, w), 3 * h + w + c), dt).requires_grad_(True)
        y = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=align)
        dy = q(fill.uniform((b, c, oh, ow), 5 * oh + ow), dt)
        y.backward(dy)
        xt = ops.nchw_to_nhwc(x.detach().cuda(), dt).t.requires_grad_(True)
        yt = ResizeBilinearFn.apply(xt, c, (oh, ow), align)
        assert rel(ops.nhwc_to_nchw(ops.NHWC(yt.detach(), c)).cpu(), y.detach()) <= TOL[dt], ("fwd", align)
        yt.backward(ops.nchw_to_nhwc(dy.cuda(), dt).t)
        assert rel(ops.nhwc_to_nchw(ops.NHWC(xt.grad, c)).cpu(), x.grad) <= 2 * TOL[dt], ("bwd", align)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", draw(10, 52))
def test_bicubic_and_nearest_forward(dt, shape):
    from climategan_amd import ops
    b, c, h, w, oh, ow = shape
    x = q(fill.uniform((b, c, h, w), 7 * h + w + c), dt)
    xg = ops.nchw_to_nhwc(x.cuda(), dt)
    ref = F.interpolate(x, size=(oh, ow), mode="bicubic", align_corners=False)
    assert rel(ops.nhwc_to_nchw(ops.resize_bicubic(xg, (oh, ow))).cpu(), ref) <= 2 * TOL[dt]
    ref = F.interpolate(x, size=(oh, ow))                                  # legacy nearest
    assert torch.equal(ops.nhwc_to_nchw(ops.resize_nearest(xg, (oh, ow))).cpu(), ref)


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", draw(10, 53))
def test_pools_forward_backward(dt, shape):
    from climategan_amd import ops
    from climategan_amd.autograd import AvgPool3x3s2Fn, MaxPool2x2Fn, MaxPool3x3s2Fn, ResizeNearest2xFn
    b, c, h, w, _, _ = shape
    h, w = max(h, 2), max(w, 2)
    # untied values so that the max-pool routing is unique: a permutation of distinct 16-bit-representable numbers
    n = b * c * h * w
    vals = (np.random.RandomState(h * 100 + w).permutation(n).astype(np.float32) - n / 2) / 64.0
    if n > 2000:
        vals = fill.uniform((n,), 9 * h + w)
    base = q(vals.reshape(b, c, h, w), dt)

    def run(fn_ref, fn_hip, even=False):
        hh, ww = (h - h % 2, w - w % 2) if even else (h, w)
        if hh < 2 or ww < 2:
            return
        x = base[:, :, :hh, :ww].clone().requires_grad_(True)
        y = fn_ref(x)
        dy = q(fill.uniform(tuple(y.shape), 17 * hh + ww), dt)
        y.backward(dy)
        xt = ops.nchw_to_nhwc(x.detach().cuda(), dt).t.requires_grad_(True)
        yt = fn_hip(xt)
        assert rel(ops.nhwc_to_nchw(ops.NHWC(yt.detach(), c)).cpu(), y.detach()) <= TOL[dt]
        yt.backward(ops.nchw_to_nhwc(dy.cuda(), dt).t)
        assert rel(ops.nhwc_to_nchw(ops.NHWC(xt.grad, c)).cpu(), x.grad) <= 2 * TOL[dt]

    run(lambda x: F.avg_pool2d(x, 3, 2, 1, count_include_pad=False), lambda t: AvgPool3x3s2Fn.apply(t, c))
    run(lambda x: F.interpolate(x, scale_factor=2), lambda t: ResizeNearest2xFn.apply(t, c))
    if n <= 2000:      # distinct values: unique arg-max
        run(lambda x: F.max_pool2d(x, 3, 2, 1), lambda t: MaxPool3x3s2Fn.apply(t, c))
        run(lambda x: F.max_pool2d(x, 2, 2), lambda t: MaxPool2x2Fn.apply(t, c), even=True)
