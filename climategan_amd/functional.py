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
