"""DeepLab-v3+ segmentation head -- mirror of the reference's ``deeplab/deeplab_v3.py`` (ResNet backbone path).

Reproduced reference behaviour: ``ConvBNReLU`` has NO ReLU (deeplab_v3.py:54-57);
``ASPPv3Plus.conv_out`` is a 1x1 conv with padding=1, so its output is (H+2)x(W+2) (deeplab_v3.py:39,90); the
decoder is called with swapped arguments ``self.decoder(z_high, z_low)`` (deeplab_v3.py:258 vs :133), i.e.
``conv_low`` runs on the ASPP output and the encoder's low-level map is the one that gets resized."""
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from ..norms import _PackCache, conv_bn_forward, needs_grad


class ConvBNReLU(nn.Module):
    """conv + BatchNorm, no ReLU (reference deeplab_v3.py:33-64)"""

    def __init__(self, in_chan, out_chan, ks=3, stride=1, padding=1, dilation=1, *args, **kwargs):
        super().__init__()
        self.conv = nn.Conv2d(in_chan, out_chan, kernel_size=ks, stride=stride, padding=padding, dilation=dilation,
                              bias=True)
        self.bn = nn.BatchNorm2d(out_chan)
        self._cache = _PackCache()
        self.init_weight()          # unconditional in the reference too (deeplab_v3.py:52), whatever ``no_init`` says

    def init_weight(self):
        """reference deeplab_v3.py:59-64: kaiming-normal with a=1 (gain 1: std = 1/sqrt(fan_in)), zero bias."""
        for ly in self.children():
            if isinstance(ly, nn.Conv2d):
                nn.init.kaiming_normal_(ly.weight, a=1)
                if ly.bias is not None:
                    nn.init.constant_(ly.bias, 0)

    def forward_nhwc(self, x, passthrough=False):
        """``passthrough``: also return x as an output of this conv's autograd node (norms.conv_bn_forward), for the next
        consumer of the same tensor."""
        if passthrough:
            return conv_bn_forward(self.conv, self.bn, self._cache, x, passthrough=True)
        return conv_bn_forward(self.conv, self.bn, self._cache, x)


class ASPPv3Plus(nn.Module):
    """reference deeplab_v3.py:67-116"""

    def __init__(self, backbone, no_init):
        super().__init__()
        if backbone != "resnet":
            raise NotImplementedError("ASPPv3Plus: only the resnet backbone has a HIP path")
        in_chan = 2048
        self.with_gp = False
        self.conv1 = ConvBNReLU(in_chan, 256, ks=1, dilation=1, padding=0)
        self.conv2 = ConvBNReLU(in_chan, 256, ks=3, dilation=6, padding=6)
        self.conv3 = ConvBNReLU(in_chan, 256, ks=3, dilation=12, padding=12)
        self.conv4 = ConvBNReLU(in_chan, 256, ks=3, dilation=18, padding=18)
        self.conv_out = ConvBNReLU(256 * 4, 256, ks=1)   # padding=1 default: output grows by 2 (reference quirk)
        if not no_init:
            self.init_weight()

    def init_weight(self):
        """reference deeplab_v3.py:111-116: loops over the DIRECT children looking for ``nn.Conv2d`` -- they are all
        ``ConvBNReLU`` blocks, so this touches nothing (each block already initialised itself); kept for the API."""
        for ly in self.children():
            if isinstance(ly, nn.Conv2d):
                nn.init.kaiming_normal_(ly.weight, a=1)
                if ly.bias is not None:
                    nn.init.constant_(ly.bias, 0)

    def forward_nhwc(self, x):
        if self.conv1.bn.training and needs_grad(self, x.t):
            # training: the four branches read x one after the other through pass-through nodes, so that their data
            # gradients are summed in the conv kernels' epilogues (autograd.ConvPassFn) instead of by three element-wise
            # passes over the 2048-channel map
            feats = []
            for c in (self.conv1, self.conv2, self.conv3):
                f, x = c.forward_nhwc(x, passthrough=True)
                feats.append(f)
            feats.append(self.conv4.forward_nhwc(x))
        else:
            feats = [c.forward_nhwc(x) for c in (self.conv1, self.conv2, self.conv3, self.conv4)]
        return self.conv_out.forward_nhwc(Fn.concat_channels(feats))


class Decoder(nn.Module):
    """reference deeplab_v3.py:119-142"""

    def __init__(self, n_classes):
        super().__init__()
        self.conv_low = ConvBNReLU(256, 48, ks=1, padding=0)
        self.conv_cat = nn.Sequential(ConvBNReLU(304, 256, ks=3, padding=1), ConvBNReLU(256, 256, ks=3, padding=1))
        self.conv_out = nn.Conv2d(256, n_classes, kernel_size=1, bias=False)
        self._cache = _PackCache()

    def forward_nhwc(self, feat_low, feat_aspp):
        h, w = feat_low.h, feat_low.w
        feat_low = self.conv_low.forward_nhwc(feat_low)
        feat_aspp_up = Fn.resize_bilinear(feat_aspp, (h, w), align_corners=True)
        feat = Fn.concat_channels([feat_low, feat_aspp_up])
        for c in self.conv_cat:
            feat = c.forward_nhwc(feat)
        return conv_bn_forward(self.conv_out, None, self._cache, feat)


class DeepLabV3Decoder(nn.Module):
    """reference deeplab_v3.py:150-271"""

    def __init__(self, opts, no_init=False, freeze_bn=False):
        super().__init__()
        num_classes = opts.gen.s.output_dim
        self.backbone = opts.gen.deeplabv3.backbone
        self.use_dada = ("d" in opts.tasks) and opts.gen.s.use_dada
        if self.backbone != "resnet":
            raise NotImplementedError("DeepLabV3Decoder: only the resnet backbone has a HIP path")
        self.aspp = ASPPv3Plus(self.backbone, no_init)
        self.decoder = Decoder(num_classes)
        from ..utils import find_target_size

        self._target_size = find_target_size(opts, "s")
        if not no_init:
            # reference deeplab_v3.py:178-190: every conv kaiming-normal on fan_out (this overrides the blocks' own
            # a=1 draw), zero biases, BatchNorm weight 1 / bias 0; then the pretrained ASPP + decoder
            for m in self.modules():
                if isinstance(m, nn.Conv2d):
                    nn.init.kaiming_normal_(m.weight, mode="fan_out")
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)
                elif isinstance(m, nn.BatchNorm2d):
                    nn.init.ones_(m.weight)
                    nn.init.zeros_(m.bias)
                elif isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, 0, 0.01)
                    nn.init.zeros_(m.bias)
            self.load_pretrained(opts)

    def load_pretrained(self, opts):
        """reference deeplab_v3.py:192-229 (resnet branch): ``aspp.*`` strictly, ``decoder.*`` minus the 19-class
        Cityscapes head, from the checkpoint ``opts.gen.deeplabv3.pretrained_model.resnet``.  The reference asserts the
        file exists; so does this -- unless the option is unset / "none" (the checkpoints are not redistributable: an
        explicit opt-out keeps the from-scratch initialisation above, and says so)."""
        from . import pretrained_path
        path = pretrained_path(opts)
        if path is None:
            print("- DeepLabV3Decoder: no pretrained_model.resnet configured, keeping the kaiming initialisation")
            return
        assert path.exists(), path
        import torch
        std = torch.load(path, map_location="cpu")
        self.aspp.load_state_dict({k.replace("aspp.", ""): v for k, v in std.items() if k.startswith("aspp.")})
        self.decoder.load_state_dict({k.replace("decoder.", ""): v for k, v in std.items()
                                      if k.startswith("decoder.") and not (len(v.shape) > 0 and v.shape[0] == 19)},
                                     strict=False)
        print("- Loaded pre-trained DeepLabv3+ (Resnet) Decoder & ASPP as Seg Decoder")

    def set_target_size(self, size):
        self._target_size = size[:2] if isinstance(size, (list, tuple)) else (size, size)

    def forward_nhwc(self, z, z_depth=None) -> ops.NHWC:
        assert isinstance(z, (tuple, list))
        if self._target_size is None:
            raise ValueError("self._target_size should be set with self.set_target_size()"
                             "to interpolate logits to the target seg map's size")
        z_high, z_low = z
        if z_depth is not None and self.use_dada:
            z_high = Fn.mul(z_high, z_depth)           # deeplab_v3.py:253-254
        z_high = self.aspp.forward_nhwc(z_high)
        s = self.decoder.forward_nhwc(z_high, z_low)            # swapped on purpose (deeplab_v3.py:258)
        ts = self._target_size if isinstance(self._target_size, (list, tuple)) else (self._target_size,) * 2
        return Fn.resize_bilinear(s, tuple(ts), align_corners=True)

    def forward(self, z, z_depth=None):
        return Fn.to_nchw(self.forward_nhwc(z, z_depth))
