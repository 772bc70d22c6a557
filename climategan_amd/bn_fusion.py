"""Mirror of the entry point of the reference's ``climategan/bn_fusion.py`` (``bn_fuse``, bn_fusion.py:97-132; called by
apply_events.py:465-466 as ``trainer.G = bn_fuse(trainer.G)`` under ``--fuse``).

The reference walks the module tree and rewrites every (Conv2d, BatchNorm2d) pair into one conv with
``w' = w * gamma / sqrt(var + eps)``, ``b' = (b - mean) * gamma / sqrt(var + eps) + beta``.  In this package that algebra is
not an optional pass: whenever a BatchNorm is in eval mode and no gradient is wanted, ``norms.conv_bn_forward`` folds it
into the PACKED weights with the same formula (``cgan_fold_bn``) and the conv kernel's epilogue does the rest -- the fp32
parameters and the state-dict layout stay those of the reference, so a fused model still saves / loads checkpoints.
``bn_fuse`` therefore has nothing left to rewrite: it puts the model in eval mode (the fusion is only valid there, and
the reference's callers are inference scripts) and returns it."""
import torch.nn as nn


def bn_fuse(model: nn.Module) -> nn.Module:
    model.eval()
    return model
