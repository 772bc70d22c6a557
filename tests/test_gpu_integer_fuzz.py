"""Seeded random-shape sweep of the bit-exact (integer / byte) kernels: min-max -> uint8 conversion, binarisation, layout
round trip with the paste, per-class counts.  Ragged sizes, single pixels, constant images (max == min), values on the
truncation boundaries."""
import numpy as np
import pytest
import torch

from climategan_amd import fill
from oracle import cpu_ref

pytestmark = pytest.mark.gpu


def shapes(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        out.append((int(rng.choice([1, 2, 5])), int(rng.choice([1, 3])), int(rng.choice([1, 2, 7, 31, 64, 129])),
                    int(rng.choice([1, 3, 8, 33, 100, 257]))))
    return out


@pytest.mark.parametrize("shape", shapes(12, 7))
def test_uint8_conversion_and_binarise_exact(shape):
    from climategan_amd import ops
    n, c, h, w = shape
    for variant in range(3):
        x = torch.from_numpy(fill.uniform((n, c, h, w), 11 * h + w + variant, -2.0, 3.0))
        if variant == 1:
            x = torch.round(x * 4) / 4          # many ties and exact boundaries (k / 255 after normalisation is rare, ties are not)
        if variant == 2:
            x[0] = 0.75                         # a constant image: max == min
        for half in (False, True):
            xx = x.half() if half else x
            ref = cpu_ref.to_uint8_hwc(xx)
            got = ops.normalize_to_uint8(xx.cuda()).cpu().numpy()
            assert got.shape == ref.shape and np.array_equal(got, ref), (shape, variant, half)
            thr = 0.5
            y, y8 = ops.binarize(xx[:, :1].contiguous().cuda(), thr, want_float=True, want_uint8=True)
            assert torch.equal(y.cpu(), (xx[:, :1] > thr).to(xx.dtype))
            assert np.array_equal(y8.cpu().numpy(), ((xx[:, :1] > thr) * 255).numpy().astype(np.uint8))


@pytest.mark.parametrize("shape", shapes(8, 8))
def test_layout_round_trip_and_paste_exact(shape):
    """NCHW fp32 -> NHWC 16-bit -> NCHW fp32 reproduces the 16-bit rounding of the input exactly; with the paste, pixels
    outside the mask come back as the fp32 originals."""
    from climategan_amd import ops
    n, _, h, w = shape
    x = torch.from_numpy(fill.uniform((n, 3, h, w), 5 * h + w)).cuda()
    m = torch.from_numpy((fill.uniform01((n, 1, h, w), 9 * h + w) > 0.5).astype(np.float32)).cuda()
    for dt in (torch.float16, torch.bfloat16):
        y = ops.nchw_to_nhwc(x, dt)
        assert y.t.shape == (n, h, w, 8) and y.t[..., 3:].abs().max().item() == 0
        back = ops.nhwc_to_nchw(y)
        assert torch.equal(back, x.to(dt).float())
        pasted = ops.nhwc_to_nchw(y, paste_x=x, paste_m=m)
        assert torch.equal(pasted, x * (1 - m) + x.to(dt).float() * m)


@pytest.mark.parametrize("shape", shapes(8, 9))
def test_seg_counts_exact(shape):
    from climategan_amd import eval_metrics, ops
    n, _, h, w = shape
    c = 11
    logits = torch.from_numpy(fill.uniform((n, c, h, w), 3 * h + w, -3, 3)).half().float()
    logits[:, 4] = logits[:, 2]                                  # exact ties: the first maximum must win
    label = torch.from_numpy((fill.uniform01((n, h, w), 13 * h + w) * (c + 2)).astype(np.int64))
    pred = logits.argmax(1)
    want = np.zeros((3, c), np.int64)
    for k in range(c):
        want[0, k] = (pred == k).sum()
        want[1, k] = (label == k).sum()
        want[2, k] = ((pred == k) & (label == k)).sum()
    for p in (logits.cuda(), ops.nchw_to_nhwc(logits.cuda(), torch.float16)):
        counts, total, _ = eval_metrics._counts(p, label.cuda())
        assert total == n * h * w and np.array_equal(counts, want)
