"""The tensor / module helpers of the reference's ``climategan/tutils.py`` that the training and inference path needs.

``init_weights`` (set-up only, host side), ``get_num_params``, ``zero_grad``, ``divide_pred`` and the two small tensor
expressions ``normalize`` / ``vgg_preprocess`` that outside callers (logger, eval scripts) use on NCHW tensors.  On the
hot path itself the last two never run as torch expressions: ``make_m_cond`` (``cgan_make_m_cond_nhwc``), the uint8
conversion (``cgan_normalize_u8_nhwc``) and the Painter heads (``cgan_painter_heads_fwd``: paste + D input +
VGG pre-processing in one kernel) have them fused.
"""
import torch
import torch.nn as nn
from torch.nn import init

from . import ops

# init_type -> how the weight of a conv / linear layer is drawn (reference tutils.py:61-76).  ``gain`` is only used by
# the three entries that take it there: xavier_uniform is called with gain 1.0 and kaiming with a=0, fan_in.
_WEIGHT_INIT = {
    "normal": lambda w, gain: init.normal_(w, 0.0, gain),
    "xavier": lambda w, gain: init.xavier_normal_(w, gain=gain),
    "xavier_uniform": lambda w, gain: init.xavier_uniform_(w, gain=1.0),
    "kaiming": lambda w, gain: init.kaiming_normal_(w, a=0, mode="fan_in"),
    "orthogonal": lambda w, gain: init.orthogonal_(w, gain=gain),
}


def init_weights(net, init_type="normal", init_gain=0.02, verbose=0, caller=""):
    """reference tutils.py:26-85.  Walks ``net.modules()`` (``net.apply``) and, keyed on the CLASS NAME like the
    reference:

    * ``*BatchNorm2d*``: weight ~ N(1, init_gain), bias = 0 (affine ones only);
    * ``*Conv*`` / ``*Linear*`` that HAVE a ``weight`` attribute: weight by ``init_type`` (normal | xavier |
      xavier_uniform | kaiming | orthogonal | none = the layer's own ``reset_parameters``), bias = 0.  A conv wrapped by
      ``SpectralNorm`` has no ``weight`` (it was replaced by ``weight_bar / weight_u / weight_v``, norms.py:123-139), so
      it is skipped -- bias included -- and ``Conv2dBlock`` itself has no ``weight``;
    * a falsy ``init_type`` / ``init_gain`` falls back to "normal" / 0.02 with the reference's notice.
    """
    who = (caller + " " + net.__class__.__name__).strip()
    if not init_type:
        print("init_weights({}): init_type is {}, defaulting to normal".format(who, init_type))
        init_type = "normal"
    if not init_gain:
        print("init_weights({}): init_gain is {}, defaulting to normal".format(who, init_type))
        init_gain = 0.02
    if init_type != "none" and init_type not in _WEIGHT_INIT:
        # the reference raises from inside apply() at the first conv it meets (tutils.py:77-80)
        if any(hasattr(m, "weight") and any(s in m.__class__.__name__ for s in ("Conv", "Linear")) and
               "BatchNorm2d" not in m.__class__.__name__ for m in net.modules()):
            raise NotImplementedError("initialization method [%s] is not implemented" % init_type)

    def visit(m):
        name = m.__class__.__name__
        if "BatchNorm2d" in name:
            if getattr(m, "weight", None) is not None:
                init.normal_(m.weight.data, 1.0, init_gain)
            if getattr(m, "bias", None) is not None:
                init.constant_(m.bias.data, 0.0)
            return
        if not hasattr(m, "weight") or not ("Conv" in name or "Linear" in name):
            return
        if init_type == "none":
            m.reset_parameters()
        else:
            _WEIGHT_INIT[init_type](m.weight.data, init_gain)
        if getattr(m, "bias", None) is not None:
            init.constant_(m.bias.data, 0.0)

    if verbose > 0:
        print("initialize %s with %s" % (net.__class__.__name__, init_type))
    net.apply(visit)
    # the draws above write through ``.data`` (as the reference does), which does not move the parameters' version
    # counters: a module re-initialised AFTER it has run a forward would keep serving its old packed weights
    # (norms._PackCache keys on ``tensor._version``)
    from . import ops
    ops.touch(*net.parameters())


def get_num_params(model):
    """reference tutils.py:411-413"""
    return sum(p.numel() for p in model.parameters())


def zero_grad(model: nn.Module):
    """reference tutils.py:430-439: gradients to None (not zero-filled)."""
    for p in model.parameters():
        p.grad = None


def normalize(t, mini=0, maxi=1):
    """reference tutils.py:567-576: min-max to [mini, maxi]; a 3-D tensor as a whole, a batch per sample -- with the
    reference's order of operations (subtract the minimum, THEN take the maximum of the shifted tensor and divide)."""
    if len(t.shape) == 3:
        return mini + (maxi - mini) * (t - t.min()) / (t.max() - t.min())
    b = t.shape[0]
    t = t - t.reshape(b, -1).min(1)[0].reshape(b, 1, 1, 1)
    t = t / t.reshape(b, -1).max(1)[0].reshape(b, 1, 1, 1)
    return mini + (maxi - mini) * t


VGG_BGR_MEAN = (103.939, 116.779, 123.680)


def vgg_preprocess(batch):
    """reference tutils.py:416-427: RGB in [-1, 1] -> BGR in [0, 255] minus the caffe channel means.  (The reference
    builds the mean tensor with a hard-coded ``.cuda()``; here it lives where ``batch`` lives.)"""
    r, g, b = torch.chunk(batch, 3, dim=1)
    bgr = (torch.cat((b, g, r), dim=1) + 1) * 255 * 0.5
    mean = torch.tensor(VGG_BGR_MEAN, dtype=bgr.dtype, device=bgr.device).reshape(1, 3, 1, 1)
    return bgr - mean


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
