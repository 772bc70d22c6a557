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
from . import autograd as ag
from .autograd import BceLogitsFn, ConvFn, HingeFn, L1Fn, MaxPool2x2Fn
from . import norms as _norms
from .norms import _PackCache


# The 16-bit type NCHW tensor arguments are converted to (the reference's call signatures hand the loss classes NCHW
# fp32 tensors; the package's own trainer hands them NHWC maps, which carry their type).  Training runs in bf16.
COMPUTE_DTYPE = torch.bfloat16


def set_compute_dtype(dtype):
    global COMPUTE_DTYPE
    if dtype not in (torch.float16, torch.bfloat16):
        raise ValueError("compute dtype must be torch.float16 or torch.bfloat16")
    COMPUTE_DTYPE = dtype


def _as_nhwc(t, what):
    """An ``ops.NHWC`` map as is; an NCHW tensor (the reference's signatures) through the layout kernel, keeping its
    graph (``autograd.FromNchwFn``)."""
    if isinstance(t, ops.NHWC):
        return t
    if torch.is_tensor(t) and t.dim() == 4:
        from . import functional as Fn
        return Fn.from_nchw(t, COMPUTE_DTYPE)
    raise TypeError("%s: expected an ops.NHWC map or an NCHW tensor, got %s" % (what, type(t).__name__))


class GANLoss(nn.Module):
    """reference losses.py:13-83: BCE-with-logits (``use_lsgan=False``, what ``get_losses`` builds, losses.py:384-388) or the
    LSGAN form (nn.MSELoss, losses.py:50-52)."""

    def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, soft_shift=0.0, flip_prob=0.0,
                 verbose=0):
        super().__init__()
        self.use_lsgan = bool(use_lsgan)          # nn.MSELoss (losses.py:50-52) instead of nn.BCEWithLogitsLoss
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
        if self.use_lsgan:
            from .autograd import MseConstFn
            return MseConstFn.apply(pred.t, pred.c, self.get_target_value(target_is_real), 1.0 / n)
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


class HingeLoss(nn.Module):
    """reference losses.py:550-593 (the Painter's GAN criterion when ``gen.p.loss == "hinge"``, losses.py:381-383):
    discriminator side ``-mean(min(x - 1, 0))`` for a real target / ``-mean(min(-x - 1, 0))`` for a fake one, generator
    side ``-mean(x)`` (which must aim for real); a list of per-scale outputs (lists: last entry) is averaged."""

    def __init__(self, tensor=torch.FloatTensor):
        super().__init__()
        self.zero_tensor = None
        self.Tensor = tensor

    def loss(self, input, target_is_real, for_discriminator=True):
        if not for_discriminator:
            assert target_is_real, "The generator's hinge loss must be aiming for real"
        pred = _as_nhwc(input, "HingeLoss")
        n = pred.n * pred.h * pred.w * pred.c
        return HingeFn.apply(pred.t, pred.c, bool(target_is_real), bool(for_discriminator), 1.0 / n)

    def __call__(self, input, target_is_real, for_discriminator=True):
        if isinstance(input, list):
            loss = 0
            for pred_i in input:
                if isinstance(pred_i, list):
                    pred_i = pred_i[-1]
                loss = loss + self.loss(pred_i, target_is_real, for_discriminator)
            return loss / len(input)
        return self.loss(input, target_is_real, for_discriminator)


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


_VGG_TAP_PASS = __import__("os").environ.get("CGAN_VGG_TAP_PASS", "1") != "0"   # same-box A/B: the taps' gradients summed by the engine
_VGG_MASK = __import__("os").environ.get("CGAN_FUSE_VGG_MASK", "1") != "0"     # same-box A/B of the VGG chain's part of norms.FUSE_RELU_MASK


class Vgg19(nn.Module):
    """reference losses.py:304-334: torchvision VGG19 ``features[0:30]`` in five slices ending at relu1_1 ... relu5_1,
    same child names (``slice1.0``, ``slice2.2``, ``slice2.5`` ...) so a torchvision state dict maps onto it.  The
    pretrained weights (``models.vgg19(pretrained=True)``) are not in the tree and cannot be downloaded here: the layers
    keep torch's default init until a state dict is loaded (VGG-loss VALUES are therefore unpinned)."""

    CFG = [(0, 3, 64), (2, 64, 64), (5, 64, 128), (7, 128, 128), (10, 128, 256), (12, 256, 256), (14, 256, 256),
           (16, 256, 256), (19, 256, 512), (21, 512, 512), (23, 512, 512), (25, 512, 512), (28, 512, 512)]
    POOLS = (4, 9, 18, 27)
    SLICES = ((0, 2), (2, 7), (7, 12), (12, 21), (21, 30))

    def __init__(self, requires_grad=False):
        super().__init__()
        convs = {i: (cin, cout) for i, cin, cout in self.CFG}
        for k, (a, b) in enumerate(self.SLICES):
            seq = nn.Sequential()
            for i in range(a, b):
                if i in convs:
                    seq.add_module(str(i), nn.Conv2d(convs[i][0], convs[i][1], 3, padding=1))
                elif i in self.POOLS:
                    seq.add_module(str(i), nn.MaxPool2d(2, 2))
                else:
                    seq.add_module(str(i), nn.ReLU(inplace=True))
            setattr(self, "slice%d" % (k + 1), seq)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False
        self._caches = {}

    def forward(self, X: ops.NHWC):
        """X: vgg_preprocess-ed NHWC map (3 channels, or the 6-channel hi | lo pair form of ``ops.painter_heads``);
        returns [relu1_1, relu2_1, relu3_1, relu4_1, relu5_1]."""
        outs = []
        y = X
        tapped = True        # y is also read by someone else (the input; a slice's result = a tap of the loss)
        for k in range(5):
            seq = getattr(self, "slice%d" % (k + 1))
            mods = list(seq.named_children())
            j = 0
            while j < len(mods):
                name, mod = mods[j]
                fusable = _norms.FUSE_RELU_MASK and _VGG_MASK and torch.is_grad_enabled() and y.t.requires_grad
                # a tapped map (relu1_1 ... relu4_1: read by the loss AND by this conv) is handed THROUGH this conv's node
                # (autograd.ConvPassFn): the loss reads the second output, its gradient comes back through this node and is
                # summed in the data-gradient kernel's epilogue -- no element-wise sum by the autograd engine, and the tap's
                # ReLU derivative can move into this layer's backward like everywhere else in the chain
                hand_through = fusable and _VGG_TAP_PASS and tapped and k > 0 and isinstance(mod, nn.Conv2d)
                # y = relu(conv(.)) read by THIS layer only: the ReLU's derivative moves into this layer's backward
                # (autograd.claim_relu_mask; norms.FUSE_RELU_MASK = False: the separate pass)
                claim = ((hand_through or not tapped) and fusable and ag.claim_relu_mask(y.t))
                tapped = False
                if isinstance(mod, nn.Conv2d):                      # conv + the ReLU that follows it, fused
                    cache = self._caches.setdefault(name, _PackCache())
                    weight = mod.weight
                    if y.c == 2 * mod.in_channels:
                        # the (hi | lo) pair form of the pre-processed image (ops.painter_heads): conv(w, hi + lo) =
                        # conv([w | w], [hi | lo]); the doubled weight is rebuilt only when the parameter changes
                        if mod.weight.requires_grad:
                            raise NotImplementedError("Vgg19: the (hi | lo) input form needs a frozen first conv")
                        key = (mod.weight._version, mod.weight.data_ptr())
                        if getattr(self, "_w2_key", None) != key:
                            self._w2, self._w2_key = torch.cat([mod.weight.data, mod.weight.data], dim=1), key
                        weight = self._w2
                    pw = cache.get((weight, mod.bias), y.t.dtype,
                                   lambda weight=weight, mod=mod: ops.pack_conv_weight(weight.data, mod.bias.data,
                                                                                       y.t.dtype))
                    if hand_through:
                        cfg = dict(c_in=y.c, stride=1, pad=1, dilation=1, act=ops.ACT_RELU, slope=0.0, mask_input=claim)
                        y_t, pass_t = ag.ConvPassFn.apply(y.t, weight, mod.bias, pw, cfg)
                        outs[-1] = ops.NHWC(pass_t, y.c)         # the loss reads the tap through this node
                        y = ops.NHWC(y_t, mod.out_channels)
                    elif torch.is_grad_enabled() and (y.t.requires_grad or mod.weight.requires_grad):
                        cfg = dict(c_in=y.c, stride=1, pad=1, dilation=1, act=ops.ACT_RELU, slope=0.0, mask_input=claim)
                        y = ops.NHWC(ConvFn.apply(y.t, weight, mod.bias, None, pw, cfg, None), mod.out_channels)
                    else:
                        y = ops.conv2d(y, pw, pad=1, act=ops.ACT_RELU)
                    j += 2
                else:                                                # max pool
                    if torch.is_grad_enabled() and y.t.requires_grad:
                        y = ops.NHWC(MaxPool2x2Fn.apply(y.t, y.c, claim), y.c)
                    else:
                        y = ops.maxpool2x2(y)
                    j += 1
            outs.append(y)
            tapped = True
        return outs


class VGGLoss(nn.Module):
    """reference losses.py:337-350: sum_i w_i L1(vgg_i(x), vgg_i(y).detach()), w = (1/32, 1/16, 1/8, 1/4, 1)."""

    def __init__(self, device=None):
        super().__init__()
        self.vgg = Vgg19().eval()
        if device is not None:
            self.vgg = self.vgg.to(device)
        self.weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

    def forward(self, x, y):
        """x, y: ``vgg_preprocess``-ed images -- NHWC maps (3 channels or the 6-channel pair form) or, as the reference
        passes them (trainer.py:1282-1284), NCHW fp32 tensors, which become pair maps here."""
        from . import functional as Fn
        if torch.is_tensor(x):
            x = Fn.from_nchw_pair(x, COMPUTE_DTYPE)
        if torch.is_tensor(y):
            y = Fn.from_nchw_pair(y.detach(), COMPUTE_DTYPE)
        x_vgg = self.vgg(_as_nhwc(x, "VGGLoss"))
        with torch.no_grad():
            y_vgg = self.vgg(_as_nhwc(y, "VGGLoss"))
        loss = 0
        for i in range(len(x_vgg)):
            f, r = x_vgg[i], y_vgg[i]
            n = f.n * f.h * f.w * f.c
            loss = loss + L1Fn.apply(f.t, r.t, f.c, self.weights[i] / n)
        return loss


# ------------------------------------------------------------------------------------------------ masker-side losses
class CrossEntropy(nn.Module):
    """reference losses.py:106-112: nn.CrossEntropyLoss(logits, target.long()); logits as an NHWC map."""

    def __call__(self, logits, target):
        logits = _as_nhwc(logits, "CrossEntropy")
        return ag.softmax_ce(logits, target.to(logits.t.device))


class TVLoss(nn.Module):
    """reference losses.py:142-169"""

    def __init__(self, tvloss_weight=1):
        super().__init__()
        self.tvloss_weight = tvloss_weight

    def forward(self, x):
        return ag.tv_loss(_as_nhwc(x, "TVLoss"), self.tvloss_weight)


class MinentLoss(nn.Module):
    """reference losses.py:172-196 (version 2 adds the variance of the entropy map)."""

    def __init__(self, version=1, lambda_var=0.1):
        super().__init__()
        self.version = version
        self.lambda_var = lambda_var

    def __call__(self, pred):
        return ag.minent_loss(_as_nhwc(pred, "MinentLoss"), self.version, self.lambda_var)


class GroundIntersectionLoss(nn.Module):
    """reference losses.py:444-450"""

    def __call__(self, pred, pseudo_ground):
        pred = _as_nhwc(pred, "GroundIntersectionLoss")
        return ag.ground_intersection(pred, pseudo_ground.to(pred.t.device))


class BCEWithLogitsLoss(nn.Module):
    """nn.BCEWithLogitsLoss as ``get_losses`` builds it for the mask (losses.py:421): 1-channel NHWC logits vs a target
    map [n, 1, h, w]."""

    def __call__(self, logits, target):
        logits = _as_nhwc(logits, "BCEWithLogitsLoss")
        return ag.bce_logits_map(logits, target.to(logits.t.device))


def prob_2_entropy(prob, depth=None):
    """reference losses.py:453-458 (``depth``: optional 1-channel NHWC map multiplied in, trainer.py:1455-1456)."""
    prob = _as_nhwc(prob, "prob_2_entropy")
    y = ag.EntropyMapFn.apply(prob.t, prob.c, depth.t if depth is not None else None)
    return ops.NHWC(y, prob.c)


def advent_input(logits, depth=None, sigmoid_pair=False):
    """prob_2_entropy(softmax(logits)) [* depth] -- or, ``sigmoid_pair``, prob_2_entropy(cat[sigmoid(x), 1 - sigmoid(x)])
    of the 1-channel mask logits -- as the (hi | lo) 16-bit pair map the ADVENT discriminators take (2C channels;
    ``FCDiscriminator`` recognises it by the channel count).  Same function of the logits as the reference's
    ``prob_2_entropy(prob)`` (losses.py:453-458, 517-519), without the 16-bit round trips through prob and entropy."""
    logits = _as_nhwc(logits, "advent_input")
    C = 2 if sigmoid_pair else logits.c
    y = ag.AdventPairFn.apply(logits.t, logits.c, bool(sigmoid_pair), depth.t if depth is not None else None)
    return ops.NHWC(y, 2 * C)


def softmax(logits):
    """torch.softmax(s, dim=1) on an NHWC map (trainer.py:1433)."""
    logits = _as_nhwc(logits, "softmax")
    return ops.NHWC(ag.SoftmaxFn.apply(logits.t, logits.c), logits.c)


def sigmoid(logits):
    """torch.sigmoid on an NHWC map, differentiable (the mask probability of trainer.py:1533)."""
    logits = _as_nhwc(logits, "sigmoid")
    return ops.NHWC(ag.SigmoidFn.apply(logits.t, logits.c), logits.c)


def sigmoid_pair(logits):
    """prob = cat[sigmoid(x), 1 - sigmoid(x)] of the mask logits (trainer.py:1533-1534) as a 2-channel NHWC map."""
    logits = _as_nhwc(logits, "sigmoid_pair")
    return ops.NHWC(ag.SigmoidPairFn.apply(logits.t), 2)


class CustomBCELoss(nn.Module):
    """reference losses.py:461-477: BCE-with-logits against a constant domain label."""

    def __call__(self, prediction, target):
        prediction = _as_nhwc(prediction, "CustomBCELoss")
        n = prediction.n * prediction.h * prediction.w * prediction.c
        return BceLogitsFn.apply(prediction.t, prediction.c, float(target), 1.0 / n)


class ADVENTAdversarialLoss(nn.Module):
    """reference losses.py:480-524.  ``gan_type="GAN"`` -> BCE (the D side, losses.py:440); anything else takes the
    reference's always-true ``elif`` and becomes the WGAN expression -mean(y D + (1 - y)(1 - D))."""

    def __init__(self, opts, gan_type="GAN"):
        super().__init__()
        self.opts = opts
        self.bce = CustomBCELoss() if gan_type == "GAN" else None

    def __call__(self, prediction, target, discriminator, depth_preds=None, logits=None, sigmoid_pair=False):
        """``logits`` (the trainer passes them): the discriminator's input is computed from the logits in fp32 and handed
        over as a 16-bit pair (``advent_input``); ``prediction`` (the probabilities) is then unused.  Without ``logits``:
        the reference's call signature, through 16-bit probability and entropy maps."""
        if self.opts.dis.m.architecture == "OmniDiscriminator":
            raise NotImplementedError("ADVENT with the OmniDiscriminator architecture has no HIP path (default: base)")
        if logits is None and torch.is_tensor(prediction):
            # the reference's signature: fp32 NCHW probabilities (and depth); the entropy is evaluated in fp32 and handed
            # to the discriminator as the same (hi | lo) pair map the logits path produces
            if isinstance(depth_preds, ops.NHWC):
                depth_preds = ops.nhwc_to_nchw(ops.detached(depth_preds))
            d_in = ops.NHWC(ag.EntropyPairFromNchwFn.apply(prediction, depth_preds, COMPUTE_DTYPE), 2 * prediction.shape[1])
        else:
            if depth_preds is not None:
                depth_preds = _as_nhwc(depth_preds, "ADVENTAdversarialLoss")
            if logits is not None:
                d_in = advent_input(logits, depth_preds, sigmoid_pair)
            else:
                d_in = prob_2_entropy(prediction, depth_preds)
        d_out = discriminator(d_in, nhwc=True)
        if self.bce is not None:
            return self.bce(d_out, target)
        return ag.advent_wgan(d_out, float(target))


class SIGMLoss(nn.Module):
    """reference losses.py:237-278: MiDaS scale-and-shift-invariant loss (medians, mean absolute deviations) plus a
    4-scale Sobel gradient-matching term; the reference's expansion of the filters to B output channels (which counts
    every response B times) is reproduced."""

    def __init__(self, gmweight=0.5, scale=4, device="cuda"):
        super().__init__()
        self.gmweight, self.scale = gmweight, scale

    def __call__(self, prediction, target):
        prediction = _as_nhwc(prediction, "SIGMLoss")
        return ag.sigm_loss(prediction, target.to(prediction.t.device), self.gmweight, self.scale)


def get_losses(opts, verbose, device=None):
    """reference losses.py:353-441: the same nested dictionary (classifier / DADA-depth options excluded)."""
    losses = {"G": {"a": {}, "p": {}, "tasks": {}}, "D": {"default": {}, "advent": {}}, "C": {}}
    if "p" in opts.tasks:
        losses["G"]["p"]["gan"] = (HingeLoss() if opts.gen.p.get("loss", "gan") == "hinge" else
                                   GANLoss(use_lsgan=False, soft_shift=opts.dis.soft_shift, flip_prob=opts.dis.flip_prob))
        losses["G"]["p"]["vgg"] = VGGLoss(device)
        losses["G"]["p"]["tv"] = TVLoss()
        losses["G"]["p"]["featmatch"] = FeatMatchLoss()
    if "d" in opts.tasks:
        losses["G"]["tasks"]["d"] = SIGMLoss(opts.train.lambdas.G.d.gml)
    if "s" in opts.tasks:
        losses["G"]["tasks"]["s"] = {"crossent": CrossEntropy(), "minent": MinentLoss(),
                                     "advent": ADVENTAdversarialLoss(opts, gan_type=opts.dis.s.gan_type)}
    if "m" in opts.tasks:
        minent = (MinentLoss(version=2, lambda_var=opts.train.lambdas.advent.ent_var) if opts.gen.m.use_minent_var
                  else MinentLoss())
        losses["G"]["tasks"]["m"] = {"bce": BCEWithLogitsLoss(), "minent": minent, "tv": TVLoss(),
                                     "advent": ADVENTAdversarialLoss(opts, gan_type=opts.dis.m.gan_type),
                                     "gi": GroundIntersectionLoss()}
    if "p" in opts.tasks:
        losses["D"]["p"] = losses["G"]["p"]["gan"]          # the SAME object: G-side calls draw smoothing / flips too
    if "m" in opts.tasks or "s" in opts.tasks:
        losses["D"]["advent"] = ADVENTAdversarialLoss(opts)
    return losses
