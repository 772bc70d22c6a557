"""Input pipeline: uint8 photo -> resize keeping the aspect ratio -> centre crop -> [-1, 1], the
pre-processing of the reference's apply_events.py:211-241 / 179-195.

The resize arithmetic lives in scikit-image 0.18.3, which is in neither the reference tree nor this image: the oracle
(``oracle.cpu_ref.resize_and_crop``) restates it around scipy's own Gaussian filter and says "parity unpinned"; the HIP
kernels are held to that restatement -- identical uint8 levels except where the float64 value sits within rounding of an
integer (then one level), exact for the cases that involve no interpolation."""
import numpy as np
import pytest
import torch

from climategan_amd import fill
from oracle import cpu_ref


def photo(h, w, seed):
    """Smooth structure + texture + saturated flats (the cases where truncation boundaries matter)."""
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = 127 + 90 * np.sin(yy / 37.0 + seed) * np.cos(xx / 53.0) + 40 * (fill.uniform01((h, w), seed) - 0.5)
    img = np.stack([base, base[::-1], base[:, ::-1]], axis=-1) + 30 * (fill.uniform01((h, w, 3), seed + 1) - 0.5)
    img[: h // 6] = 255.0
    img[-h // 8:, : w // 3] = 0.0
    return np.clip(img, 0, 255).astype(np.uint8)


def test_oracle_geometry_and_identity():
    img = photo(64, 64, 1)
    out = cpu_ref.resize_and_crop(img, to=64)               # scale 1: no filter, no interpolation
    assert np.array_equal((out * 255).round().astype(np.uint8), img)
    out = cpu_ref.resize_and_crop(photo(90, 150, 2), to=48)
    assert out.shape == (48, 48, 3) and out.min() >= 0 and out.max() <= 1
    x = cpu_ref.to_m1_p1(out)
    assert x.dtype == np.float32 and x.min() >= -1 and x.max() <= 1
    with pytest.raises(ValueError):
        cpu_ref.to_m1_p1(out * 300)
    flat = np.full((100, 140, 3), 200, np.uint8)            # a flat image stays flat up to one truncation level
    o = (cpu_ref.resize_and_crop(flat, to=32) * 255).round()
    assert o.min() >= 199 and o.max() <= 200


def test_oracle_resize_agrees_with_scipy_on_both_halves():
    """What CAN be pinned of the scikit-image restatement without scikit-image (round 6): both halves of ``resize`` against
    scipy.ndimage, the library skimage itself is built on and the only resampling code installed here --
    (i) the anti-aliasing pre-filter IS skimage's own call (``ndi.gaussian_filter(image, (s - 1) / 2, mode='mirror')``,
        transform/_warps.py); a scale of 1 along an axis must leave that axis untouched (sigma 0);
    (ii) the warp (order 1, mode 'reflect', pixel centres at +0.5) against ``ndi.map_coordinates(order=1, mode='mirror')`` --
        the routine skimage's own ``warp`` falls back to outside its 2-D fast path -- on the same sampling coordinates:
        float64 agreement to 1e-9 on enlarging, shrinking and mixed cases, borders included.
    Still not a pin against scikit-image's bytes (its 2-D fast path is Cython of its own): DESIGN 3 keeps N3 'unpinned'."""
    from scipy import ndimage as ndi

    for (h, w), (rows, cols) in (((90, 150), (48, 80)), ((61, 47), (128, 99)), ((120, 80), (120, 33)), ((33, 70), (95, 70))):
        img = photo(h, w, h + w).astype(np.float64)
        got = cpu_ref.skimage_resize_018(img, (rows, cols))
        fr, fc = h / rows, w / cols
        sig = (max(0.0, (fr - 1) / 2), max(0.0, (fc - 1) / 2), 0.0)
        blurred = ndi.gaussian_filter(img, sig, cval=0, mode="mirror")
        if fr <= 1 and fc <= 1:
            assert np.array_equal(blurred, img)                              # enlarging: no pre-filter at all
        r = fr * (np.arange(rows) + 0.5) - 0.5
        c = fc * (np.arange(cols) + 0.5) - 0.5
        rr, cc = np.meshgrid(r, c, indexing="ij")
        ref = np.stack([ndi.map_coordinates(blurred[..., k], [rr, cc], order=1, mode="mirror") for k in range(3)], axis=-1)
        ref = np.clip(ref, blurred.min(), blurred.max())
        assert got.shape == (rows, cols, 3)
        assert np.abs(got - ref).max() <= 1e-9 * 255, (h, w, rows, cols, np.abs(got - ref).max())


def test_geometry_matches_reference_formula():
    from climategan_amd import ops
    for h, w, to in ((480, 640, 640), (1000, 750, 640), (640, 640, 640), (333, 517, 128), (2000, 3000, 640)):
        rows, cols, top, left = ops.resize_crop_geometry(h, w, to)
        size = (to, int(to * w / h)) if h < w else (int(to * h / w), to)      # apply_events.py:224-228
        assert (rows, cols) == size
        assert (top, left) == ((rows - to) // 2, (cols - to) // 2)            # apply_events.py:236-237


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,to", [(700, 900, 640), (1300, 1000, 640), (640, 640, 640), (500, 400, 640),
                                    (961, 1283, 320), (640, 1500, 640)])
def test_resize_and_crop_matches_oracle(h, w, to):
    from climategan_amd import apply_events
    img = photo(h, w, h + w)
    ref = cpu_ref.to_m1_p1(cpu_ref.resize_and_crop(img, to))                   # HWC float32
    got = apply_events.resize_and_crop(img, to).cpu().numpy().transpose(1, 2, 0)
    assert got.shape == ref.shape == (to, to, 3)
    lv_ref = np.round((ref / 2 + 0.5) * 255).astype(int)
    lv_got = np.round((got / 2 + 0.5) * 255).astype(int)
    diff = np.abs(lv_ref - lv_got)
    assert diff.max() <= 1, diff.max()
    assert (diff != 0).mean() <= 1e-4, (diff != 0).mean()
    same = diff == 0
    assert np.array_equal(got[same], ref[same])             # the fp32 [-1, 1] values themselves are bit-identical
    if h == w == to:
        assert diff.max() == 0


@pytest.mark.gpu
def test_prepare_batch_feeds_infer_all_layout():
    from climategan_amd import apply_events
    imgs = [photo(700, 900, 3), photo(800, 650, 4)]
    b = apply_events.prepare_batch(imgs, to=256)
    assert b.shape == (2, 3, 256, 256) and b.dtype == torch.float32 and b.is_cuda
    for i, im in enumerate(imgs):
        one = apply_events.resize_and_crop(im, 256)
        assert torch.equal(b[i], one)
    with pytest.raises(ValueError):
        apply_events.prepare_batch([np.zeros((10, 10, 4), np.uint8)], to=8)
    with pytest.raises(ValueError):
        apply_events.resize_and_crop(np.zeros((10, 10, 3), np.float32), 8)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,max_w", [(700, 900, -1), (1300, 1000, 640), (512, 768, -1), (400, 650, 1024)])
def test_resize_keep_ratio_matches_oracle(h, w, max_w):
    """apply_events' keep_ratio branch: multiples of 128, no crop, no uint8 truncation (float image)."""
    from climategan_amd import apply_events
    img = photo(h, w, 3 * h + w)
    ref = cpu_ref.resize_keep_ratio(img, max_w)                                # HWC float32
    got = apply_events.resize_keep_ratio(img, max_w).cpu().numpy().transpose(1, 2, 0)
    nh, nw = apply_events.to_128(img, max_w)
    assert got.shape == ref.shape == (nh, nw, 3) and nh % 128 == 0 and nw % 128 == 0
    assert np.abs(got - ref).max() <= 3e-7                                      # float32 rounding of values in [-1, 1]
