"""The tensor helpers of the reference's ``climategan/tutils.py`` that the training path needs."""
from . import ops


def _half(t, first):
    if isinstance(t, ops.NHWC):
        n = t.t.shape[0] // 2
        return ops.NHWC(t.t[:n] if first else t.t[n:], t.c)
    n = t.size(0) // 2
    return t[:n] if first else t[n:]


def divide_pred(disc_output):
    """reference tutils.py:443-469: split a discriminator output computed on a batch-concatenation of two sets
    (real ‖ fake) into the two halves; works on NCHW tensors and on ``ops.NHWC`` maps (batch is the leading
    dimension of both layouts, so the halves are contiguous views)."""
    if type(disc_output) == list:
        half1 = [[_half(t, True) for t in p] for p in disc_output]
        half2 = [[_half(t, False) for t in p] for p in disc_output]
        return half1, half2
    return _half(disc_output, True), _half(disc_output, False)
