"""Host-side mirror of the reference's ``climategan/deeplab`` package (DeepLab-v3+ / ResNet-101 path only; the
MobileNetV2 and DeepLab-v2 alternatives are not in the benchmark configs)."""
import torch.nn as nn

from .deeplab_v3 import DeepLabV3Decoder
from .resnet101_v3 import ResNet101


def create_encoder(opts, no_init=False, verbose=0):
    """reference deeplab/__init__.py:12-25"""
    if opts.gen.encoder.architecture == "deeplabv3":
        return build_v3_backbone(opts, no_init)
    raise NotImplementedError("Unknown encoder: {} (only deeplabv3 has a HIP path)".format(opts.gen.encoder.architecture))


def create_segmentation_decoder(opts, no_init=False, verbose=0):
    """reference deeplab/__init__.py:28-40"""
    if opts.gen.s.architecture == "deeplabv3":
        return DeepLabV3Decoder(opts, no_init)
    raise NotImplementedError("Unknown Segmentation architecture: {}".format(opts.gen.s.architecture))


def build_v3_backbone(opts, no_init, verbose=0):
    """reference deeplab/__init__.py:43-101 (pretrained-weight loading is the caller's job here: the checkpoints are
    not redistributable; ``load_state_dict`` accepts the reference's keys)."""
    if opts.gen.deeplabv3.backbone != "resnet":
        raise NotImplementedError("deeplabv3 backbone '%s' has no HIP path (resnet only)" % opts.gen.deeplabv3.backbone)
    return ResNet101(output_stride=opts.gen.deeplabv3.output_stride, BatchNorm=nn.BatchNorm2d, verbose=verbose,
                     no_init=no_init)
