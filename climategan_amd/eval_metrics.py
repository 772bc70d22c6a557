"""Host-side mirror of the reference's ``climategan/eval_metrics.py`` validation metrics used by
``Trainer.eval_images`` (trainer.py:1706-1790): ``accuracy`` (eval_metrics.py:67-76) and ``mIOU`` (eval_metrics.py:79-130).

The reference moves predictions to the CPU and loops over classes with ``.item()`` synchronisations; here one HIP kernel
(``cgan_seg_counts``) produces the per-class counts (predicted / labelled / both) in one pass over the device tensors
and the ratios are formed from 3 x C integers.  Integer work: results are exactly the reference's for the same logits.
"""

import numpy as np
import torch

from . import _lib, ops


def _counts(pred, label):
    """pred: ``ops.NHWC`` logits or an [N, C, H, W] device tensor; label: [N, H, W] or [N, 1, H, W]."""
    if isinstance(pred, ops.NHWC):
        n, hw, c, layout, dtype, p = pred.n, pred.h * pred.w, pred.c, 0, pred.dtype_id, pred.t
        if pred.cs != ops.cs8(c):
            raise RuntimeError("metrics: NHWC logits must be stored with round_up(c, 8) channels")
    else:
        if pred.dim() != 4:
            raise ValueError("metrics: [N, C, H, W] logits expected, got %s" % (tuple(pred.shape),))
        p = pred.contiguous().float()
        n, c, layout, dtype = p.shape[0], p.shape[1], 1, 0
        hw = p.shape[2] * p.shape[3]
    ops._need_cuda(p, label)
    if label.dim() == 4:
        assert label.shape[1] == 1                                    # eval_metrics.py:70-72
        label = label[:, 0]
    if label.numel() != n * hw:
        raise ValueError("metrics: %d labels for %d predictions" % (label.numel(), n * hw))
    lab = label.contiguous().float()
    counts = torch.zeros((3, c), dtype=torch.int64, device=p.device)
    lib = _lib.load()
    _lib.check(lib.cgan_seg_counts(ops._ptr(p), layout, dtype, n, hw, c, ops._ptr(lab), ops._ptr(counts), ops._stream()),
               "cgan_seg_counts")
    return counts.cpu().numpy(), n * hw, lab


def accuracy(pred_im, gt_im):
    """eval_metrics.py:67-76: ``(argmax_c(pred) == gt).sum() / gt.size``.

    Reproduced as the reference computes it, quirks included: the label array is taken BEFORE its channel axis is
    squeezed, so (i) with [N, 1, H, W] labels and N > 1 the comparison broadcasts across samples -- refused here, pass
    [N, H, W] labels or one sample at a time as ``Trainer.eval_images`` does; (ii) a 1-channel prediction (the binarised
    mask of trainer.py:1763-1768) is arg-maxed to all zeros, so its "accuracy" is the fraction of zero labels."""
    n = pred_im.n if isinstance(pred_im, ops.NHWC) else pred_im.shape[0]
    c = pred_im.c if isinstance(pred_im, ops.NHWC) else pred_im.shape[1]
    if gt_im.dim() == 4 and n > 1:
        raise NotImplementedError("accuracy: [N, 1, H, W] labels with N > 1 broadcast across samples in the reference "
                                  "(eval_metrics.py:68-76); pass [N, H, W] labels or single samples")
    if c == 1:
        if isinstance(pred_im, ops.NHWC):
            raise NotImplementedError("accuracy: 1-channel NHWC predictions are not used by the reference's callers")
        pred_im = torch.cat([pred_im.float() * 0 + 1, pred_im.float() * 0], dim=1)     # argmax == 0 everywhere
    counts, total, _ = _counts(pred_im, gt_im)
    return float(counts[2].sum()) / total


def mIOU(pred, label, average="macro"):
    """eval_metrics.py:79-130 (pred: logits; label: integer class map).  With 2 classes only ``label.max()`` is scored;
    classes absent from both prediction and label are skipped; nan when nothing is left."""
    counts, _, lab = _counts(pred, label)
    num_classes = counts.shape[1]
    interesting = list(range(num_classes)) if num_classes > 2 else [int(lab.max().item())]
    weights, ious = [], []
    for k in interesting:
        n_pred = int(counts[0][k]) if 0 <= k < num_classes else 0
        n_tgt = int(counts[1][k]) if 0 <= k < num_classes else 0
        if n_tgt > 0 or n_pred > 0:
            inter = int(counts[2][k])
            weights.append(n_pred)
            ious.append(float(inter) / float(n_pred + n_tgt - inter))
    if not ious:
        return float("nan")
    if average == "weighted":
        return np.sum(np.multiply(weights, ious) / np.sum(weights))
    return np.mean(ious)
