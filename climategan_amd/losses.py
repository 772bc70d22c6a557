"""Host-side mirror of the GAN-side losses of the reference's ``climategan/losses.py``:
``GANLoss`` and ``FeatMatchLoss``, same constructor arguments and call signatures, evaluated by the HIP loss kernels
(value and gradient in one pass) on the NHWC feature maps a ``climategan_amd`` discriminator returns with
``nhwc=True``.  The RNG-dependent parts (one-sided label smoothing, label flipping) draw from the same host RNGs as
the reference (Python ``random`` and ``torch.FloatTensor.uniform_``), so seeded runs see the same targets.
"""
from random import random as rand

import torch
import torch.nn as nn

from . import ops
from .autograd import BceLogitsFn, L1Fn


def _as_nhwc(t, what):
    if not isinstance(t, ops.NHWC):
        raise TypeError("%s: expected the NHWC maps of a climategan_amd discriminator called with nhwc=True, got %s"
                        % (what, type(t).__name__))
    return t


class GANLoss(nn.Module):
    """reference losses.py:13-83 (BCE-with-logits form, ``use_lsgan=False`` as built by ``get_losses``,
    losses.py:384-388)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, soft_shift=0.0, flip_prob=0.0,
                 verbose=0):
        super().__init__()
        if use_lsgan:
            raise NotImplementedError("GANLoss: the LSGAN (MSE) form has no HIP kernel (the reference's get_losses "
                                      "builds use_lsgan=False, losses.py:384-388)")
        self.soft_shift = soft_shift
        self.verbose = verbose
        self.register_buffer("real_label", torch.tensor(target_real_label))
        self.register_buffer("fake_label", torch.tensor(target_fake_label))
        self.flip_prob = flip_prob
        self._real, self._fake = float(target_real_label), float(target_fake_label)

    def get_target_value(self, target_is_real):
        """losses.py:56-64: one scalar soft_change per call, real - change or fake + change."""
        soft_change = float(torch.FloatTensor(1).uniform_(0, self.soft_shift))   # drawn even when soft_shift == 0
        return self._real - soft_change if target_is_real else self._fake + soft_change

    def _one(self, pred, target_is_real):
        pred = _as_nhwc(pred, "GANLoss")
        n = pred.n * pred.h * pred.w * pred.c
        return BceLogitsFn.apply(pred.t, pred.c, self.get_target_value(target_is_real), 1.0 / n)

    def __call__(self, input, target_is_real, *args, **kwargs):
        r = rand()
        if isinstance(input, list):
            loss = 0
            for pred_i in input:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                if r < self.flip_prob:
                    target_is_real = not target_is_real          # toggles per scale, as the reference does (:73-74)
                loss = loss + self._one(pred_i, target_is_real)
            return loss / len(input)
        if r < self.flip_prob:
            target_is_real = not target_is_real
        return self._one(input, target_is_real)


class FeatMatchLoss(nn.Module):
    """reference losses.py:86-103: sum over discriminators and intermediate layers of L1(fake, real.detach()) / num_D."""

    def __call__(self, pred_real, pred_fake):
        num_D = len(pred_fake)
        total = 0.0
        for i in range(num_D):
            for j in range(len(pred_fake[i]) - 1):
                f, r = _as_nhwc(pred_fake[i][j], "FeatMatchLoss"), _as_nhwc(pred_real[i][j], "FeatMatchLoss")
                n = f.n * f.h * f.w * f.c
                total = total + L1Fn.apply(f.t, r.t.detach(), f.c, 1.0 / (n * num_D))
        return total
