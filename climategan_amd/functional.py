"""Grad-aware wrappers: each picks the plain HIP op under no_grad and the matching ``autograd`` Function (HIP forward
and backward) when the input carries a gradient, so module ``forward_nhwc`` code reads the same in both modes."""
import torch

from . import ops


def _tracked(*xs) -> bool:
    return torch.is_grad_enabled() and any(x is not None and x.t.requires_grad for x in xs)


def resize_bilinear(x: ops.NHWC, size, align_corners=False) -> ops.NHWC:
    if _tracked(x):
        from .autograd import ResizeBilinearFn
        return ops.NHWC(ResizeBilinearFn.apply(x.t, x.c, tuple(int(s) for s in size), bool(align_corners)), x.c)
    return ops.resize_bilinear(x, size, align_corners=align_corners)


def upsample_nearest2x(x: ops.NHWC) -> ops.NHWC:
    if _tracked(x):
        from .autograd import ResizeNearest2xFn
        return ops.NHWC(ResizeNearest2xFn.apply(x.t, x.c), x.c)
    return ops.resize_nearest(x, (x.h * 2, x.w * 2))


def maxpool3x3s2(x: ops.NHWC) -> ops.NHWC:
    if _tracked(x):
        from .autograd import MaxPool3x3s2Fn
        return ops.NHWC(MaxPool3x3s2Fn.apply(x.t, x.c), x.c)
    return ops.maxpool3x3s2(x)


def concat_channels(xs) -> ops.NHWC:
    if _tracked(*xs):
        from .autograd import ConcatFn
        cs = [x.c for x in xs]
        return ops.NHWC(ConcatFn.apply(cs, *[x.t for x in xs]), sum(cs))
    return ops.concat_channels(xs)


def mul(a: ops.NHWC, b: ops.NHWC) -> ops.NHWC:
    if _tracked(a, b):
        from .autograd import MulFn
        return ops.NHWC(MulFn.apply(a.t, b.t, a.c), a.c)
    return ops.eltwise_mul(a, b)


def add_act(a: ops.NHWC, b: ops.NHWC, act=ops.ACT_NONE, slope=0.2) -> ops.NHWC:
    from .autograd import AddActFn
    return ops.NHWC(AddActFn.apply(a.t, b.t, a.c, act, slope), a.c)


def to_nchw(y: ops.NHWC, paste_x=None, paste_m=None) -> torch.Tensor:
    """fp32 NCHW tensor of an NHWC map (optionally pasted: x (1 - m) + y m), with its graph when ``y`` carries one."""
    if _tracked(y):
        from .autograd import ToNchwFn
        return ToNchwFn.apply(y.t, y.c, paste_x, paste_m)
    return ops.nhwc_to_nchw(y, paste_x, paste_m)


def from_nchw(x, dtype, cs=None, mask=None) -> ops.NHWC:
    """16-bit NHWC map of an NCHW tensor (optionally times (1 - mask)); differentiable w.r.t. ``x``.  An ``ops.NHWC``
    argument is handed through."""
    if isinstance(x, (ops.NHWC, ops.PairMap)):
        return x
    if torch.is_grad_enabled() and x.requires_grad:
        from .autograd import FromNchwFn
        return ops.NHWC(FromNchwFn.apply(x, dtype, cs, mask), x.shape[1])
    return ops.nchw_to_nhwc(x, dtype, cs=cs, mask=mask)


def from_nchw_pair(x, dtype) -> ops.NHWC:
    """(hi | lo) 16-bit pair map of an fp32 NCHW tensor (2C channels, hi + lo = x to ~16 bits of mantissa)."""
    from .autograd import FromNchwPairFn
    if torch.is_grad_enabled() and x.requires_grad:
        return ops.NHWC(FromNchwPairFn.apply(x, dtype), 2 * x.shape[1])
    with torch.no_grad():
        return ops.NHWC(FromNchwPairFn.apply(x.detach(), dtype), 2 * x.shape[1])


def resize_nearest(x: ops.NHWC, size, cs_out=None) -> ops.NHWC:
    if _tracked(x):
        from .autograd import ResizeNearestFn
        return ops.NHWC(ResizeNearestFn.apply(x.t, x.c, tuple(int(s) for s in size), cs_out), x.c)
    return ops.resize_nearest(x, size, cs_out=cs_out)


def sigmoid(x: ops.NHWC) -> ops.NHWC:
    if _tracked(x):
        from .autograd import SigmoidFn
        return ops.NHWC(SigmoidFn.apply(x.t, x.c), x.c)
    return ops.sigmoid(x)


def resize_bicubic(x: ops.NHWC, size) -> ops.NHWC:
    if _tracked(x):
        from .autograd import ResizeBicubicFn
        return ops.NHWC(ResizeBicubicFn.apply(x.t, x.c, tuple(int(s) for s in size)), x.c)
    return ops.resize_bicubic(x, size)
