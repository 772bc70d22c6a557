"""ResNet-101 (output stride 8/16, dilated) encoder -- mirror of the reference's ``deeplab/resnet101_v3.py``.

Same module tree / state-dict keys (conv1, bn1, layer{1..4}.{i}.{conv1,bn1,conv2,bn2,conv3,bn3,downsample.{0,1}}).
HIP forward: every conv is the MFMA implicit-GEMM kernel with the eval-mode BatchNorm folded into its weights and
ReLU / residual add in its epilogue; 3x3/s2 max-pool kernel; NHWC 16-bit throughout."""
import torch.nn as nn

from .. import functional as Fn
from .. import ops
from ..norms import DEFAULT_COMPUTE_DTYPE, _PackCache, conv_bn_forward, needs_grad


FUSE_RESIDUAL_GRADIENT = True      # A/B switch for tools / tests: False = the engine's own gradient accumulation


class Bottleneck(nn.Module):
    """reference resnet101_v3.py:4-50"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, BatchNorm=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = BatchNorm(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=stride, dilation=dilation, padding=dilation,
                               bias=False)
        self.bn2 = BatchNorm(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = BatchNorm(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        self.dilation = dilation
        self._c = [_PackCache() for _ in range(4)]

    def forward_nhwc(self, x, sole_consumer=False):
        """``sole_consumer``: the caller vouches that nothing but this block reads ``x`` (the previous block's output) -- the
        derivative of the ReLU that produced it then rides in conv1's data-gradient kernel (norms.conv_bn_forward)."""
        residual = x
        if FUSE_RESIDUAL_GRADIENT and self.bn1.training and needs_grad(self, x.t):
            # training: conv1 hands x through to the residual branch (the identity, or the downsample conv), so that x has
            # ONE consumer in the autograd graph and the two gradient contributions are summed inside conv1's
            # data-gradient kernel (autograd.ConvPassFn)
            out, residual = conv_bn_forward(self.conv1, self.bn1, self._c[0], x, act=ops.ACT_RELU, passthrough=True,
                                            sole_consumer=sole_consumer)
            x = residual
        else:
            out = conv_bn_forward(self.conv1, self.bn1, self._c[0], x, act=ops.ACT_RELU)
        out = conv_bn_forward(self.conv2, self.bn2, self._c[1], out, act=ops.ACT_RELU)
        if self.downsample is not None:
            residual = conv_bn_forward(self.downsample[0], self.downsample[1], self._c[3], x)
        return conv_bn_forward(self.conv3, self.bn3, self._c[2], out, act=ops.ACT_RELU, residual=residual)


class ResNet(nn.Module):
    """reference resnet101_v3.py:53-187"""

    def __init__(self, block, layers, output_stride, BatchNorm, verbose=0, no_init=False):
        self.inplanes = 64
        self.verbose = verbose
        super().__init__()
        blocks = [1, 2, 4]
        if output_stride == 16:
            strides, dilations = [1, 2, 2, 1], [1, 1, 1, 2]
        elif output_stride == 8:
            strides, dilations = [1, 2, 1, 1], [1, 1, 2, 4]
        else:
            raise NotImplementedError
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0], strides[0], dilations[0], BatchNorm)
        self.layer2 = self._make_layer(block, 128, layers[1], strides[1], dilations[1], BatchNorm)
        self.layer3 = self._make_layer(block, 256, layers[2], strides[2], dilations[2], BatchNorm)
        self.layer4 = self._make_MG_unit(block, 512, blocks, strides[3], dilations[3], BatchNorm)
        self.compute_dtype = DEFAULT_COMPUTE_DTYPE
        self.pair_precision = False
        self._stem = _PackCache()

    def _downsample(self, block, planes, stride, BatchNorm):
        if stride != 1 or self.inplanes != planes * block.expansion:
            return nn.Sequential(nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                                 BatchNorm(planes * block.expansion))
        return None

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1, BatchNorm=None):
        layers = [block(self.inplanes, planes, stride, dilation, self._downsample(block, planes, stride, BatchNorm), BatchNorm)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, dilation=dilation, BatchNorm=BatchNorm))
        return nn.Sequential(*layers)

    def _make_MG_unit(self, block, planes, blocks, stride=1, dilation=1, BatchNorm=None):
        layers = [block(self.inplanes, planes, stride, dilation=blocks[0] * dilation,
                        downsample=self._downsample(block, planes, stride, BatchNorm), BatchNorm=BatchNorm)]
        self.inplanes = planes * block.expansion
        for i in range(1, len(blocks)):
            layers.append(block(self.inplanes, planes, stride=1, dilation=blocks[i] * dilation, BatchNorm=BatchNorm))
        return nn.Sequential(*layers)

    def forward_nhwc(self, x: ops.NHWC):
        x = conv_bn_forward(self.conv1, self.bn1, self._stem, x, act=ops.ACT_RELU)
        x = Fn.maxpool3x3s2(x)
        # a block's output is read by the next block only -- except the max-pool's (no bottleneck behind it), layer1's last
        # (``low`` also feeds the decoders) and the encoder's result
        sole = False
        for b in self.layer1:
            x = b.forward_nhwc(x, sole_consumer=sole)
            sole = True
        low = x
        sole = False
        for layer in (self.layer2, self.layer3, self.layer4):
            for b in layer:
                x = b.forward_nhwc(x, sole_consumer=sole)
                sole = True
        return x, low

    def forward(self, input):
        """NCHW in; returns (z_high, z_low) as NHWC containers (consumed by the decoders of this package)."""
        if getattr(self, "pair_precision", False):
            # split-precision inference (G.set_compute_dtype("pair16")): every activation of the Masker as hi + lo
            import torch
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and self.training:
                raise NotImplementedError("pair16 is an inference mode (eval(), no_grad())")
            return self.forward_nhwc(ops.pair_from_nchw(input, self.compute_dtype))
        x = Fn.from_nchw(input, self.compute_dtype)
        return self.forward_nhwc(x)


def ResNet101(output_stride=8, BatchNorm=nn.BatchNorm2d, verbose=0, no_init=False):
    """reference resnet101_v3.py:190-203"""
    return ResNet(Bottleneck, [3, 4, 23, 3], output_stride, BatchNorm, verbose=verbose, no_init=no_init)
