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


def pretrained_path(opts):
    """``opts.gen.deeplabv3.pretrained_model.resnet`` as a Path, or None when it is unset / empty / "none" (the
    reference's default is a cluster path, defaults.yaml:117-120; ``config.default_opts`` leaves it out)."""
    from pathlib import Path
    try:
        p = opts.gen.deeplabv3.pretrained_model.resnet
    except (AttributeError, KeyError, TypeError):
        return None
    if not p or not isinstance(p, (str, Path)) or str(p).lower() == "none":
        return None
    return Path(p)


def build_v3_backbone(opts, no_init, verbose=0):
    """reference deeplab/__init__.py:43-67: ResNet-101 with torch's default conv / BatchNorm initialisation (the
    reference's ResNet has no init code of its own), then -- unless ``no_init`` -- the ``backbone.*`` entries of the
    pretrained DeepLab-v3+ checkpoint, which must exist when it is configured."""
    if opts.gen.deeplabv3.backbone != "resnet":
        raise NotImplementedError("deeplabv3 backbone '%s' has no HIP path (resnet only)" % opts.gen.deeplabv3.backbone)
    resnet = ResNet101(output_stride=opts.gen.deeplabv3.output_stride, BatchNorm=nn.BatchNorm2d, verbose=verbose,
                       no_init=no_init)
    if not no_init:
        path = pretrained_path(opts)
        if path is None:
            print("    - ResNet101 encoder: no pretrained_model.resnet configured, keeping torch's default init")
        else:
            assert path.exists(), path
            import torch
            std = torch.load(path, map_location="cpu")
            resnet.load_state_dict({k.replace("backbone.", ""): v for k, v in std.items() if k.startswith("backbone.")})
            print("    - Loaded pre-trained DeepLabv3+ Resnet101 Backbone as Encoder")
    return resnet
