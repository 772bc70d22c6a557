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
