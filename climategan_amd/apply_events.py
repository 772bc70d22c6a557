"""Host-side mirror of the data half of the reference's ``apply_events.py``: the
pre-processing in front of ``Trainer.infer_all`` -- ``resize_and_crop`` (apply_events.py:211-241) and ``to_m1_p1``
(apply_events.py:179-195) -- as one HIP call per image, writing straight into the batch tensor the Masker reads.

The reference does this on the CPU with scikit-image (``resize(..., anti_aliasing=True)``), ~25 ms per 1-2 Mpixel photo;
here the uint8 image is uploaded once (3 bytes / pixel) and everything else happens on the device.
"""
import torch

from . import ops


def resize_and_crop(img, to=640, device="cuda"):
    """uint8 HWC image (numpy array or tensor) -> fp32 [3, to, to] device tensor in [-1, 1]
    (= ``to_m1_p1(resize_and_crop(img, to))`` transposed to CHW).  RGBA must be converted by the caller as the
    reference does (apply_events.py:491)."""
    t = img if isinstance(img, torch.Tensor) else torch.from_numpy(img)
    if t.dtype != torch.uint8:
        raise ValueError("resize_and_crop: np.uint8 255 image expected (apply_events.py:218), got %s" % t.dtype)
    return ops.resize_and_crop_u8(t.to(device), to)


def prepare_batch(images, to=640, device="cuda"):
    """list of uint8 HWC images of any sizes -> fp32 [B, 3, to, to] in [-1, 1] (the ``np.stack(images)`` the reference
    hands to ``infer_all``, apply_events.py:521-524, already NCHW and on the device)."""
    if len(images) == 0:
        raise ValueError("prepare_batch: no images")
    batch = torch.empty((len(images), 3, to, to), dtype=torch.float32, device=device)
    for i, img in enumerate(images):
        t = img if isinstance(img, torch.Tensor) else torch.from_numpy(img)
        if t.dim() != 3 or t.shape[2] != 3:
            raise ValueError("prepare_batch: image %d is not [H, W, 3] (convert RGBA / grey first)" % i)
        ops.resize_and_crop_u8(t.to(device), to, out=batch[i])
    return batch


def to_128(im, w_target=-1):
    """reference utils.py:998-1007: (nh, nw) = the largest multiples of 128 not above w_target and nw * h / w."""
    h, w = im.shape[:2]
    aspect_ratio = h / w
    if w_target < 0:
        w_target = w
    nw = int(w_target / 128) * 128
    nh = int(nw * aspect_ratio / 128) * 128
    return nh, nw


def resize_keep_ratio(img, max_im_width=-1, device="cuda"):
    """The keep_ratio branch of the reference loop (apply_events.py:494-497, 502): uint8 HWC image -> fp32
    [3, nh, nw] in [-1, 1] with (nh, nw) = to_128(img, max_im_width); images of different sizes cannot be stacked, so
    ``infer_all`` takes them one at a time (batch_size 1, as the reference requires for keep_ratio)."""
    t = img if isinstance(img, torch.Tensor) else torch.from_numpy(img)
    if t.dtype != torch.uint8:
        raise ValueError("resize_keep_ratio: np.uint8 255 image expected, got %s" % t.dtype)
    nh, nw = to_128(t, max_im_width)
    if nh <= 0 or nw <= 0:
        raise ValueError("resize_keep_ratio: image %s is smaller than 128 pixels in one direction" % (tuple(t.shape),))
    return ops.resize_u8(t.to(device), (nh, nw))
