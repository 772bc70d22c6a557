"""CPU restatement of the ClimateGAN generator/discriminator hot path (TEST INFRASTRUCTURE).

Own code, functional style, keyed on the reference's state-dict layout; every function cites the reference
file:line it follows (paths relative to the reference repo root).  Runs in torch fp32 (default) or fp64 on
the host.  Used as (a) the parity checker for the HIP path in ``tests/`` and ``__graft_entry__.smoke()``
and (b) the reported ``cpu_baseline`` in ``bench.py`` -- never on the product path.

Pinned by: tests/test_oracle_golden.py (committed vectors produced by the real reference) and
tests/test_oracle_vs_reference.py (direct comparison with the imported reference, dev container only).
Parity UNPINNED for arithmetic that lives in third-party libraries which are neither in the reference tree nor in this
image: the three library formulas inside ``add_fire`` (kornia's Gaussian kernel / ``filter2d``, torchvision's
``adjust_contrast`` / ``adjust_brightness`` -- everything ELSE in ``add_fire`` is pinned by the reference's own fire.py run
with those three bound to their documented formulas, golden ``fire_small``), ``skimage_resize_018`` (scikit-image 0.18.3
``resize`` behind apply_events.resize_and_crop) and the pretrained VGG-19 WEIGHTS (the VGG loss itself is pinned by the
reference's Vgg19 / VGGLoss classes with portable-fill weights, golden ``vgg_small``); DESIGN.md section 3.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def sub(sd: SD, prefix: str) -> SD:
    """Sub-state-dict with ``prefix`` stripped."""
    p = prefix + "."
    return {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}


# --------------------------------------------------------------------------------------------------
# norms.py
# --------------------------------------------------------------------------------------------------
def l2normalize(v: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """climategan/norms.py:80-81"""
    return v / (v.norm() + eps)


def spectral_norm_step(w_bar: torch.Tensor, u: torch.Tensor, v: torch.Tensor, power_iterations: int = 1):
    """One ``SpectralNorm._update_u_v`` (climategan/norms.py:100-112).

    Returns (w, u_new, v_new, sigma).  Runs on EVERY forward in the reference, eval included
    (norms.py:141-143); callers must write u_new/v_new back into the state.
    """
    h = w_bar.shape[0]
    wm = w_bar.reshape(h, -1)
    wd = wm.detach()                       # the reference iterates on ``.data`` (norms.py:103-106): no graph
    for _ in range(power_iterations):
        v = l2normalize(torch.mv(wd.t(), u))
        u = l2normalize(torch.mv(wd, v))
    sigma = u.dot(wm.mv(v))                # autograd reaches w_bar here and in w_bar / sigma only (norms.py:107-112)
    return w_bar / sigma, u, v, sigma


def sn_conv2d(x, sd: SD, prefix: str, stride=1, padding=0, dilation=1, update: bool = True):
    """``SpectralNorm(nn.Conv2d).forward`` (norms.py:141-143): power-iterate, then convolve with w_bar/sigma.

    Mutates ``sd[prefix.module.weight_u/_v]`` in place when ``update`` (as the reference does).
    """
    wb = sd[prefix + ".module.weight_bar"]
    u = sd[prefix + ".module.weight_u"]
    v = sd[prefix + ".module.weight_v"]
    w, u2, v2, _ = spectral_norm_step(wb, u, v)
    if update:
        sd[prefix + ".module.weight_u"] = u2
        sd[prefix + ".module.weight_v"] = v2
    b = sd.get(prefix + ".module.bias")
    return F.conv2d(x, w, b, stride=stride, padding=padding, dilation=dilation)


def instance_norm(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.InstanceNorm2d(affine=False, track_running_stats=False): biased var over H*W (norms.py:151)."""
    mean = x.mean(dim=(2, 3), keepdim=True)
    var = x.var(dim=(2, 3), unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps)


def batch_norm_param_free(x: torch.Tensor, running_mean=None, running_var=None, training=True, eps=1e-5):
    """nn.BatchNorm2d(affine=False) (norms.py:155): batch stats in train, running stats in eval."""
    if training or running_mean is None:
        mean = x.mean(dim=(0, 2, 3), keepdim=True)
        var = x.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    else:
        mean = running_mean.view(1, -1, 1, 1)
        var = running_var.view(1, -1, 1, 1)
    return (x - mean) / torch.sqrt(var + eps)


def nearest_resize(x: torch.Tensor, size: Tuple[int, int]) -> torch.Tensor:
    """F.interpolate(mode='nearest') legacy rule src = floor(dst * in / out) (norms.py:179, painter.py:152)."""
    n, c, h, w = x.shape
    oh, ow = size
    if (oh, ow) == (h, w):
        return x
    iy = torch.floor(torch.arange(oh, dtype=torch.float64) * (h / oh)).long().clamp_(max=h - 1)
    ix = torch.floor(torch.arange(ow, dtype=torch.float64) * (w / ow)).long().clamp_(max=w - 1)
    return x[:, :, iy][:, :, :, ix]


def spade(x: torch.Tensor, segmap: torch.Tensor, sd: SD, prefix: str, norm_type: str = "instance",
          training: bool = True) -> torch.Tensor:
    """``SPADE.forward`` (climategan/norms.py:174-186); hidden width 128 (norms.py:163)."""
    if norm_type == "instance":
        normalized = instance_norm(x)
    elif norm_type == "batch":
        normalized = batch_norm_param_free(
            x, sd.get(prefix + ".param_free_norm.running_mean"), sd.get(prefix + ".param_free_norm.running_var"),
            training)
    else:
        raise ValueError("%s is not a recognized param-free norm type in SPADE" % norm_type)
    seg = nearest_resize(segmap, x.shape[2:])
    w0, b0 = sd[prefix + ".mlp_shared.0.weight"], sd[prefix + ".mlp_shared.0.bias"]
    pw = w0.shape[-1] // 2
    actv = F.relu(F.conv2d(seg, w0, b0, padding=pw))
    gamma = F.conv2d(actv, sd[prefix + ".mlp_gamma.weight"], sd[prefix + ".mlp_gamma.bias"], padding=pw)
    beta = F.conv2d(actv, sd[prefix + ".mlp_beta.weight"], sd[prefix + ".mlp_beta.bias"], padding=pw)
    return normalized * (1 + gamma) + beta


def _conv_maybe_sn(x, sd: SD, prefix: str, padding: int, update: bool):
    if prefix + ".module.weight_bar" in sd:
        return sn_conv2d(x, sd, prefix, padding=padding, update=update)
    return F.conv2d(x, sd[prefix + ".weight"], sd.get(prefix + ".bias"), padding=padding)


def spade_resnet_block(x, seg, sd: SD, prefix: str, norm_type="instance", last_activation=None,
                       training=True, update=True):
    """``SPADEResnetBlock.forward`` (climategan/blocks.py:369-395).

    Order of spectral-norm updates follows the reference: shortcut (conv_s) first, then conv_0, conv_1.
    """
    learned_shortcut = (prefix + ".conv_s.module.weight_bar" in sd) or (prefix + ".conv_s.weight" in sd)
    if learned_shortcut:
        x_s = _conv_maybe_sn(spade(x, seg, sd, prefix + ".norm_s", norm_type, training), sd, prefix + ".conv_s",
                             0, update)
    else:
        x_s = x
    dx = _conv_maybe_sn(F.leaky_relu(spade(x, seg, sd, prefix + ".norm_0", norm_type, training), 0.2), sd,
                        prefix + ".conv_0", 1, update)
    dx = _conv_maybe_sn(F.leaky_relu(spade(dx, seg, sd, prefix + ".norm_1", norm_type, training), 0.2), sd,
                        prefix + ".conv_1", 1, update)
    out = x_s + dx
    if last_activation == "lrelu":
        return F.leaky_relu(out, 0.2)
    if last_activation is None:
        return out
    raise NotImplementedError("The type of activation is not supported: {}".format(last_activation))


# --------------------------------------------------------------------------------------------------
# painter.py / generator.py (paint)
# --------------------------------------------------------------------------------------------------
def painter_blocks(sd: SD) -> List[str]:
    n_up = len({k.split(".")[1] for k in sd if k.startswith("up_spades.")})
    return ["head_0", "G_middle_0", "G_middle_1"] + ["up_spades.%d" % i for i in range(n_up)] + ["final_spade"]


def painter_forward(sd: SD, cond: torch.Tensor, z_h: int, z_w: int, z: Optional[torch.Tensor] = None,
                    update: bool = True) -> torch.Tensor:
    """``PainterSpadeDecoder.forward`` (climategan/painter.py:149-168), use_final_shortcut=False.

    ``sd`` is the painter state dict (keys as in the reference, no ``painter.`` prefix).  Spectral-norm
    u/v entries are updated in place (one power iteration per wrapped conv, reference norms.py:100-112).
    """
    if z is None:
        z = F.conv2d(nearest_resize(cond, (z_h, z_w)), sd["fc.weight"], sd["fc.bias"], padding=1)
    up = lambda t: nearest_resize(t, (t.shape[-2] * 2, t.shape[-1] * 2))  # blocks.py:28-43
    y = spade_resnet_block(z, cond, sd, "head_0", update=update)
    y = up(y)
    y = spade_resnet_block(y, cond, sd, "G_middle_0", update=update)
    y = up(y)
    y = spade_resnet_block(y, cond, sd, "G_middle_1", update=update)
    n_up = len({k.split(".")[1] for k in sd if k.startswith("up_spades.")})
    for i in range(n_up):
        y = up(y)
        y = spade_resnet_block(y, cond, sd, "up_spades.%d" % i, update=update)
    y = spade_resnet_block(y, cond, sd, "final_spade", update=update)
    y = F.conv2d(F.leaky_relu(y, 0.2), sd["conv_img.weight"], sd["conv_img.bias"], padding=1)
    return torch.tanh(y)


def paint(sd: SD, m: torch.Tensor, x: torch.Tensor, z_h: int, z_w: int, no_paste: bool = False,
          paste_original_content: bool = True, update: bool = True) -> torch.Tensor:
    """``OmniGenerator.paint`` (climategan/generator.py:279-297) with ``no_z: true`` (z=None)."""
    m = m.to(x.dtype)
    fake = painter_forward(sd, x * (1.0 - m), z_h, z_w, update=update)
    if paste_original_content and not no_paste:
        return x * (1.0 - m) + fake * m
    return fake


# --------------------------------------------------------------------------------------------------
# discriminator.py
# --------------------------------------------------------------------------------------------------
def nlayer_discriminator(x, sd: SD, prefix: str, n_layers: int = 4, update=True) -> List[torch.Tensor]:
    """``NLayerDiscriminator.forward`` (climategan/discriminator.py:172-182) with instance norm,
    get_intermediate_features=True: returns n_layers+2 tensors."""
    p = prefix + "." if prefix else ""
    outs = []
    # model0: SN conv 4x4 s2 + LeakyReLU (discriminator.py:100-109)
    y = F.leaky_relu(sn_conv2d(x, sd, p + "model0.0", stride=2, padding=1, update=update), 0.2)
    outs.append(y)
    for n in range(1, n_layers):  # discriminator.py:113-134
        y = sn_conv2d(y, sd, p + "model%d.0" % n, stride=2, padding=1, update=update)
        y = F.leaky_relu(instance_norm(y), 0.2)
        outs.append(y)
    y = sn_conv2d(y, sd, p + "model%d.0" % n_layers, stride=1, padding=1, update=update)  # :138-154
    y = F.leaky_relu(instance_norm(y), 0.2)
    outs.append(y)
    y = sn_conv2d(y, sd, p + "model%d.0" % (n_layers + 1), stride=1, padding=1, update=update)  # :157-163
    outs.append(y)
    return outs


def multiscale_discriminator(x, sd: SD, num_D: int = 3, n_layers: int = 4, update=True):
    """``MultiscaleDiscriminator.forward`` (climategan/discriminator.py:227-239)."""
    result = []
    for i in range(num_D):
        result.append(nlayer_discriminator(x, sd, "discriminator_%d" % i, n_layers, update))
        x = F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False)  # :223-225
    return result


def fc_discriminator(x, sd: SD, prefix: str = "", update=True):
    """``get_fc_discriminator(use_norm=True)`` (climategan/discriminator.py:327-349)."""
    p = prefix + "." if prefix else ""
    for i, idx in enumerate((0, 2, 4, 6, 8)):
        key = p + str(idx)
        if key + ".module.weight_bar" in sd:
            x = sn_conv2d(x, sd, key, stride=2, padding=1, update=update)
        else:
            x = F.conv2d(x, sd[key + ".weight"], sd[key + ".bias"], stride=2, padding=1)
        if i < 4:
            x = F.leaky_relu(x, 0.2)
    return x


# --------------------------------------------------------------------------------------------------
# optim.py: ExtraAdam
# --------------------------------------------------------------------------------------------------
class ExtraAdamRef:
    """Functional restatement of ``ExtraAdam`` (climategan/optim.py:200-291) + ``Extragradient.extrapolation/step``
    (optim.py:153-197) on a list of tensors.  ``grads[i] is None`` skips the parameter like the reference."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = params
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.state = [dict(step=0, exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p)) for p in params]
        self.params_copy = []

    def _update(self, i, grad):
        import math

        if grad is None:
            return None
        st = self.state[i]
        b1, b2 = self.betas
        st["step"] += 1
        if self.wd != 0:
            grad = grad + self.wd * self.params[i]
        st["exp_avg"] = st["exp_avg"] * b1 + (1 - b1) * grad
        st["exp_avg_sq"] = st["exp_avg_sq"] * b2 + (1 - b2) * grad * grad
        denom = st["exp_avg_sq"].sqrt() + self.eps
        step_size = self.lr * math.sqrt(1 - b2 ** st["step"]) / (1 - b1 ** st["step"])
        return -step_size * st["exp_avg"] / denom

    def extrapolation(self, grads):
        is_empty = len(self.params_copy) == 0
        for i, g in enumerate(grads):
            u = self._update(i, g)
            if is_empty:
                self.params_copy.append(self.params[i].clone())
            if u is not None:
                self.params[i] = self.params[i] + u

    def step(self, grads):
        if len(self.params_copy) == 0:
            raise RuntimeError("Need to call extrapolation before calling step.")
        for i, g in enumerate(grads):
            u = self._update(i, g)
            if u is not None:
                self.params[i] = self.params_copy[i] + u
        self.params_copy = []


# --------------------------------------------------------------------------------------------------
# Masker (inference / eval-mode BatchNorm): deeplab/resnet101_v3.py, deeplab/deeplab_v3.py, depth.py, blocks.py
# --------------------------------------------------------------------------------------------------
_BN_TRAINING = [False]


class bn_training:
    """``with bn_training():`` -- every ``_bn`` inside behaves like ``nn.BatchNorm2d`` in TRAINING mode (batch
    statistics, running statistics moved with momentum 0.1 into the state dict, ``num_batches_tracked`` + 1): the Masker
    under ``G.train()`` (reference trainer.py:933).  Default (outside): eval mode."""

    def __enter__(self):
        self.prev = _BN_TRAINING[0]
        _BN_TRAINING[0] = True

    def __exit__(self, *a):
        _BN_TRAINING[0] = self.prev


def _bn(x, sd: SD, prefix: str, eps: float = 1e-5):
    """nn.BatchNorm2d: eval mode, or training mode inside ``bn_training()``"""
    if _BN_TRAINING[0]:
        rm, rv = sd[prefix + ".running_mean"].clone(), sd[prefix + ".running_var"].clone()
        y = F.batch_norm(x, rm, rv, sd.get(prefix + ".weight"), sd.get(prefix + ".bias"), True, 0.1, eps)
        sd[prefix + ".running_mean"], sd[prefix + ".running_var"] = rm, rv
        if prefix + ".num_batches_tracked" in sd:
            sd[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
        return y
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd.get(prefix + ".weight"),
                        sd.get(prefix + ".bias"), False, 0.0, eps)


def bottleneck(x, sd: SD, p: str, stride: int, dilation: int):
    """``Bottleneck.forward`` (deeplab/resnet101_v3.py:30-50)"""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
    out = F.relu(_bn(F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=dilation, dilation=dilation), sd,
                     p + ".bn2"))
    out = _bn(F.conv2d(out, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    residual = x
    if p + ".downsample.0.weight" in sd:
        residual = _bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
    return F.relu(out + residual)


def resnet101(x, sd: SD, prefix: str = "", output_stride: int = 8):
    """``ResNet.forward`` (deeplab/resnet101_v3.py:176-187), layers [3,4,23,3], MG unit [1,2,4] (:135-174)."""
    p = prefix + "." if prefix else ""
    strides, dilations = ([1, 2, 1, 1], [1, 1, 2, 4]) if output_stride == 8 else ([1, 2, 2, 1], [1, 1, 1, 2])
    x = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], stride=2, padding=3), sd, p + "bn1"))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    low = None
    for li, nblocks in enumerate([3, 4, 23, 3]):
        for b in range(nblocks):
            if li < 3:
                dil = dilations[li]
            else:
                dil = [1, 2, 4][b] * dilations[3]
            x = bottleneck(x, sd, "%slayer%d.%d" % (p, li + 1, b), strides[li] if b == 0 else 1, dil)
        if li == 0:
            low = x
    return x, low


def _conv_bn(x, sd: SD, p: str, padding=0, dilation=1):
    """``ConvBNReLU.forward`` = conv + BN, NO ReLU (deeplab/deeplab_v3.py:54-57)"""
    return _bn(F.conv2d(x, sd[p + ".conv.weight"], sd[p + ".conv.bias"], padding=padding, dilation=dilation), sd,
               p + ".bn")


def deeplab_v3_decoder(z, sd: SD, prefix: str, target_size, z_depth=None, use_dada=True):
    """``DeepLabV3Decoder.forward`` (deeplab_v3.py:244-266) incl. ASPPv3Plus (:95-109) and Decoder (:133-142) with
    the reference's swapped decoder arguments and the padding=1 1x1 ``conv_out``."""
    p = prefix + "." if prefix else ""
    z_high, z_low = z
    if z_depth is not None and use_dada:
        z_high = z_high * z_depth
    a = p + "aspp."
    feat = torch.cat([_conv_bn(z_high, sd, a + "conv1"), _conv_bn(z_high, sd, a + "conv2", 6, 6),
                      _conv_bn(z_high, sd, a + "conv3", 12, 12), _conv_bn(z_high, sd, a + "conv4", 18, 18)], 1)
    aspp = _conv_bn(feat, sd, a + "conv_out", padding=1)           # 1x1 conv with padding 1: (H+2) x (W+2)
    d = p + "decoder."
    feat_low, feat_aspp = aspp, z_low                              # decoder(z_high, z_low): arguments swapped
    h, w = feat_low.shape[2:]
    fl = _conv_bn(feat_low, sd, d + "conv_low")
    fa = F.interpolate(feat_aspp, (h, w), mode="bilinear", align_corners=True)
    f = torch.cat([fl, fa], 1)
    f = _conv_bn(f, sd, d + "conv_cat.0", 1)
    f = _conv_bn(f, sd, d + "conv_cat.1", 1)
    logits = F.conv2d(f, sd[d + "conv_out.weight"])
    return F.interpolate(logits, size=target_size, mode="bilinear", align_corners=True)


def conv2d_block(x, sd: SD, p: str, k: int, padding: int, pad_type: str, norm: str, activ: str, update=True):
    """``Conv2dBlock.forward`` (blocks.py:138-143) for the configurations on the Masker path."""
    if padding > 0:
        x = F.pad(x, (padding,) * 4, mode="reflect" if pad_type == "reflect" else "constant")
    if p + ".conv.module.weight_bar" in sd:
        x = sn_conv2d(x, sd, p + ".conv", update=update)
    else:
        x = F.conv2d(x, sd[p + ".conv.weight"], sd.get(p + ".conv.bias"))
    if norm in ("batch", "spectral_batch"):
        x = _bn(x, sd, p + ".norm")
    if activ == "lrelu":
        x = F.leaky_relu(x, 0.2)
    elif activ == "relu":
        x = F.relu(x)
    return x


def dada_depth_decoder(z, sd: SD, prefix: str, target_size: Optional[int] = None):
    """``DADADepthDecoder.forward`` (depth.py:128-155), feature fusion on, upsample_featuremaps on.  ``target_size``
    is the int the ctor stores (depth.py:105-107): when the depth map's width differs, it is re-sampled bicubically
    to 384x384 and then nearest to (target, target) (depth.py:143-153); None = no re-sampling."""
    p = prefix + "." if prefix else ""
    zz = z[0] if isinstance(z, (tuple, list)) else z
    e = conv2d_block(zz, sd, p + "enc4_1", 1, 0, "reflect", "batch", "lrelu")
    e = conv2d_block(e, sd, p + "enc4_2", 3, 1, "reflect", "batch", "lrelu")
    e = conv2d_block(e, sd, p + "enc4_3", 1, 0, "reflect", "batch", "lrelu")
    z_depth = conv2d_block(e, sd, p + "dec4", 1, 0, "zero", "none", "lrelu")
    u = nearest_resize(e, (e.shape[2] * 2, e.shape[3] * 2))
    u = conv2d_block(u, sd, p + "upsample.1", 3, 1, "reflect", "batch", "lrelu")
    u = F.conv2d(u, sd[p + "upsample.2.weight"], sd[p + "upsample.2.bias"])
    depth = torch.mean(u, dim=1, keepdim=True)
    if target_size is not None and depth.shape[-1] != target_size:
        depth = F.interpolate(depth, size=(384, 384), mode="bicubic", align_corners=False)
        depth = F.interpolate(depth, (target_size, target_size), mode="nearest")
    return depth, z_depth


def mask_base_decoder(z, sd: SD, prefix: str, n_res=3, n_upsample=3, update=True):
    """``BaseDecoder.forward`` (blocks.py:292-313) as configured for the mask (masker.py:25-56): spectral norm,
    LeakyReLU, reflect padding, low-level features, no DADA fusion."""
    p = prefix + "." if prefix else ""
    zz, low = z
    low = conv2d_block(low, sd, p + "low_level_conv", 3, 1, "reflect", "spectral", "lrelu", update)
    low = F.interpolate(low, size=zz.shape[-2:], mode="bilinear")
    zz = conv2d_block(zz, sd, p + "proj_conv", 1, 0, "zero", "spectral", "lrelu", update)
    zz = conv2d_block(torch.cat([low, zz], 1), sd, p + "merge_feats_conv", 1, 0, "reflect", "spectral", "lrelu", update)
    for r in range(n_res):
        q = "%smodel.0.model.%d.model." % (p, r)
        out = conv2d_block(zz, sd, q + "0", 3, 1, "reflect", "spectral", "lrelu", update)
        out = conv2d_block(out, sd, q + "1", 3, 1, "reflect", "spectral", "none", update)
        zz = out + zz
    for i in range(n_upsample):
        zz = nearest_resize(zz, (zz.shape[2] * 2, zz.shape[3] * 2))
        zz = conv2d_block(zz, sd, "%smodel.%d" % (p, 2 + 2 * i), 3, 1, "reflect", "spectral", "lrelu", update)
    return conv2d_block(zz, sd, "%smodel.%d" % (p, 1 + 2 * n_upsample), 3, 1, "reflect", "none", "none", update)


def masker_forward(sd: SD, x: torch.Tensor, s_target, update=True, d_target: Optional[int] = None):
    """Masker inference as in ``Trainer.infer_all`` (trainer.py:272-287) with the default config:
    z = encode(x); d, z_depth = dec_d(z); s = dec_s(z, z_depth); m = sigmoid(dec_m(z))."""
    z = resnet101(x, sd, "encoder")
    d, z_depth = dada_depth_decoder(z, sd, "decoders.d", d_target)
    s = deeplab_v3_decoder(z, sd, "decoders.s", s_target, z_depth, use_dada=True)
    m = torch.sigmoid(mask_base_decoder(z, sd, "decoders.m", update=update))
    return {"d": d, "s": s, "m": m, "z_high": z[0], "z_depth": z_depth}


# --------------------------------------------------------------------------------------------------
# Inference harness: Trainer.infer_all / compute_flood (climategan/trainer.py:217-334, 1844-1877)
# --------------------------------------------------------------------------------------------------
def normalize(t: torch.Tensor, mini=0, maxi=1) -> torch.Tensor:
    """``tutils.normalize`` (tutils.py:567-576): per-image min-max to [mini, maxi]."""
    if t.dim() == 3:
        return mini + (maxi - mini) * (t - t.min()) / (t.max() - t.min())
    b = t.shape[0]
    t = t - t.reshape(b, -1).min(1)[0].reshape(b, 1, 1, 1)
    t = t / t.reshape(b, -1).max(1)[0].reshape(b, 1, 1, 1)
    return mini + (maxi - mini) * t


def to_uint8_hwc(t: torch.Tensor):
    """trainer.py:311-326: normalize -> permute to NHWC -> numpy -> ``(a * 255).astype(uint8)`` (truncation)."""
    a = normalize(t).permute(0, 2, 3, 1).numpy()
    return (a * 255).astype("uint8")


def compute_flood(sd: SD, x: torch.Tensor, m: torch.Tensor, z_h: int, z_w: int, bin_value: float = -1,
                  update: bool = True) -> torch.Tensor:
    """``Trainer.compute_flood`` (trainer.py:1844-1877) with a given mask, non-cloudy: optional binarisation then
    ``G.paint(m, x)``.  ``sd`` = generator state dict (painter keys prefixed ``painter.``)."""
    if bin_value >= 0:
        m = (m > bin_value).to(m.dtype)
    return paint(sub(sd, "painter"), m, x, z_h, z_w, update=update)


def infer_all_flood(sd: SD, x: torch.Tensor, n_up: int, bin_value: float = -1, s_target=(160, 160), d_target=160):
    """``Trainer.infer_all`` (trainer.py:217-334), flood event only: masker stages in the reference's order, flood,
    uint8 conversion, and the uint8 mask of ``return_masks``."""
    z_h, z_w = x.shape[-2] // 2 ** n_up, x.shape[-1] // 2 ** n_up          # painter.set_latent_shape(x.shape, True)
    mk = masker_forward(sd, x, s_target, d_target=d_target)
    flood = compute_flood(sd, x, mk["m"], z_h, z_w, bin_value)
    return {"flood": flood, "flood_u8": to_uint8_hwc(flood), "m": mk["m"], "d": mk["d"], "s": mk["s"],
            "mask_u8": ((mk["m"] > bin_value) * 255).numpy().astype("uint8")}


# --------------------------------------------------------------------------------------------------
# Training: Painter discriminator update (climategan/trainer.py:1073-1107) with GANLoss (losses.py:13-83)
# --------------------------------------------------------------------------------------------------
def gan_loss(preds, target_is_real: bool, real_label=1.0, fake_label=0.0) -> torch.Tensor:
    """``GANLoss.__call__`` with use_lsgan=False, soft_shift=0, flip_prob=0: mean over scales of
    BCEWithLogits(pred_i[-1], target)."""
    loss = 0
    for p in preds:
        p = p[-1] if isinstance(p, (list, tuple)) else p
        t = torch.full_like(p, real_label if target_is_real else fake_label)
        loss = loss + F.binary_cross_entropy_with_logits(p, t)
    return loss / len(preds)


def hinge_loss(preds, target_is_real: bool, for_discriminator: bool = True) -> torch.Tensor:
    """``HingeLoss.__call__`` (losses.py:550-593): per scale (last entry of a list) -mean(min(+-x - 1, 0)) on the
    discriminator side, -mean(x) on the generator side; mean over scales."""
    if not isinstance(preds, (list, tuple)):
        preds = [preds]
    loss = 0
    for p in preds:
        p = p[-1] if isinstance(p, (list, tuple)) else p
        if for_discriminator:
            v = (p - 1) if target_is_real else (-p - 1)
            loss = loss - torch.minimum(v, torch.zeros_like(v)).mean()
        else:
            assert target_is_real, "The generator's hinge loss must be aiming for real"
            loss = loss - p.mean()
    return loss / len(preds)


def painter_d_step(sd_d: SD, m: torch.Tensor, x: torch.Tensor, fake: torch.Tensor, num_D: int, n_layers: int):
    """D-side painter loss and parameter gradients: D(cat_batch[cat_ch(m, x), cat_ch(m, fake)]) -> divide ->
    GANLoss(fake, False) + GANLoss(real, True); returns (loss, {key: grad}).  ``sd_d`` holds D["p"]'s tensors."""
    params = {k: v.clone().requires_grad_(k.endswith("weight_bar") or k.endswith("bias")) for k, v in sd_d.items()}
    real_cat = torch.cat([m, x], dim=1)
    fake_cat = torch.cat([m, fake], dim=1)
    out = multiscale_discriminator(torch.cat([real_cat, fake_cat], dim=0), params, num_D, n_layers)
    real_d = [[t[: t.size(0) // 2] for t in p] for p in out]
    fake_d = [[t[t.size(0) // 2:] for t in p] for p in out]
    loss = gan_loss(fake_d, False) + gan_loss(real_d, True)
    keys = [k for k, v in params.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [params[k] for k in keys])
    for k in sd_d:                                    # power-iterated u / v back into the caller's state
        if k.endswith("weight_u") or k.endswith("weight_v"):
            sd_d[k] = params[k].detach()
    return loss.detach(), dict(zip(keys, grads))


def feat_match_loss(pred_real, pred_fake) -> torch.Tensor:
    """``FeatMatchLoss.__call__`` (losses.py:86-103)."""
    num_D = len(pred_fake)
    loss = 0.0
    for i in range(num_D):
        for j in range(len(pred_fake[i]) - 1):
            loss = loss + F.l1_loss(pred_fake[i][j], pred_real[i][j].detach()) / num_D
    return loss


def painter_g_step(sd_p: SD, sd_d: SD, m: torch.Tensor, x: torch.Tensor, z_h: int, z_w: int, num_D: int, n_layers: int,
                   lambda_featmatch: float = 10.0):
    """G-side painter loss of ``get_painter_loss`` (trainer.py:1256-1387, single-discriminator branch, VGG / TV /
    context / reconstruction off): fake = paint(m, x); D on cat_batch[cat_ch(m, x), cat_ch(m, fake)];
    GANLoss(fake_d, True) (unscaled, trainer.py:1369-1371) + lambda * FeatMatch(real_d, fake_d); gradients w.r.t. the
    Painter's trainable tensors (D frozen).  Returns (loss, {key: grad}, {term: value})."""
    pp = {k: v.clone().requires_grad_(not (k.endswith("weight_u") or k.endswith("weight_v"))) for k, v in sd_p.items()}
    dd = {k: v.clone() for k, v in sd_d.items()}
    fake = paint(pp, m, x, z_h, z_w)
    real_cat = torch.cat([m, x], dim=1)
    fake_cat = torch.cat([m, fake], dim=1)
    out = multiscale_discriminator(torch.cat([real_cat, fake_cat], dim=0), dd, num_D, n_layers)
    real_d = [[t[: t.size(0) // 2] for t in p] for p in out]
    fake_d = [[t[t.size(0) // 2:] for t in p] for p in out]
    gan = gan_loss(fake_d, True)
    fm = feat_match_loss(real_d, fake_d) * lambda_featmatch
    loss = gan + fm
    keys = [k for k, v in pp.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [pp[k] for k in keys])
    return loss.detach(), dict(zip(keys, grads)), {"gan": gan.detach(), "featmatch": fm.detach()}


# --------------------------------------------------------------------------------------------------
# VGG loss: losses.py:304-350 (Vgg19 slices + VGGLoss), tutils.py:416-427 (vgg_preprocess)
# --------------------------------------------------------------------------------------------------
# torchvision's published configuration "E" (VGG-19) up to features[29] (relu5_1): the reference takes
# ``models.vgg19(pretrained=True).features`` and cuts it into slices [0,2) [2,7) [7,12) [12,21) [21,30) (losses.py:313-322).
# The ARCHITECTURE is pinned through the reference's own Vgg19 / VGGLoss classes (golden ``vgg_small``); the pretrained
# WEIGHTS are a download that is not in the tree, so loss VALUES with the real weights stay unpinned.
VGG19_E = (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512)
VGG_SLICE_ENDS = (2, 7, 12, 21, 30)
VGG_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)


def vgg19_layers():
    """[(feature index, "conv", cin, cout) | (index, "relu") | (index, "pool")] for features[0:30]."""
    layers, i, cin = [], 0, 3
    for v in VGG19_E:
        if v == "M":
            layers.append((i, "pool"))
            i += 1
        else:
            layers += [(i, "conv", cin, v), (i + 1, "relu")]
            cin = v
            i += 2
    return [l for l in layers if l[0] < 30]


def vgg19_shapes() -> Dict[str, Tuple[int, ...]]:
    """State-dict layout of the reference's ``Vgg19`` module: ``slice<k>.<feature index>.{weight,bias}``."""
    out = {}
    for l in vgg19_layers():
        if l[1] == "conv":
            k = next(j for j, e in enumerate(VGG_SLICE_ENDS) if l[0] < e) + 1
            out["slice%d.%d.weight" % (k, l[0])] = (l[3], l[2], 3, 3)
            out["slice%d.%d.bias" % (k, l[0])] = (l[3],)
    return out


def vgg_preprocess(batch: torch.Tensor) -> torch.Tensor:
    """tutils.py:416-427: RGB -> BGR, [-1, 1] -> [0, 255], minus the caffe means (103.939, 116.779, 123.680)."""
    r, g, b = torch.chunk(batch, 3, dim=1)
    bgr = (torch.cat((b, g, r), dim=1) + 1) * 255 * 0.5
    return bgr - torch.tensor([103.939, 116.779, 123.680], dtype=bgr.dtype).reshape(1, 3, 1, 1)


def vgg19_features(x: torch.Tensor, sd: SD) -> List[torch.Tensor]:
    """``Vgg19.forward`` (losses.py:326-334): [relu1_1, relu2_1, relu3_1, relu4_1, relu5_1]."""
    outs, k = [], 1
    for l in vgg19_layers():
        if l[0] == VGG_SLICE_ENDS[k - 1]:
            outs.append(x)
            k += 1
        if l[1] == "conv":
            p = "slice%d.%d" % (k, l[0])
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=1)
        elif l[1] == "relu":
            x = F.relu(x)
        else:
            x = F.max_pool2d(x, 2, 2)
    outs.append(x)
    return outs


def vgg_loss(sd: SD, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """``VGGLoss.forward`` (losses.py:345-350): sum_i w_i * L1(vgg_i(x), vgg_i(y).detach())."""
    fx, fy = vgg19_features(x, sd), vgg19_features(y, sd)
    return sum(w * F.l1_loss(a, b.detach()) for w, a, b in zip(VGG_WEIGHTS, fx, fy))


def painter_vgg_term(sd_vgg: SD, fake: torch.Tensor, x: torch.Tensor, m: torch.Tensor, lambda_vgg: float = 10.0):
    """The VGG term of ``get_painter_loss`` (trainer.py:1276-1287) on a painter output ``fake`` BEFORE the paste:
    fake_flooded = x (1 - m) + fake m (generator.py:295-296); lambda * VGGLoss(pre(fake_flooded * m), pre(x * m))."""
    fake_flooded = x * (1.0 - m) + fake * m
    return vgg_loss(sd_vgg, vgg_preprocess(fake_flooded * m), vgg_preprocess(x * m)) * lambda_vgg


# --------------------------------------------------------------------------------------------------
# Smog event: Trainer.compute_smog (climategan/trainer.py:1879-1939), tutils.srgb2lrgb / lrgb2srgb (:534-565)
# --------------------------------------------------------------------------------------------------
def srgb2lrgb(x: torch.Tensor) -> torch.Tensor:
    x = normalize(x)
    im = ((x + 0.055) / 1.055) ** 2.4
    return torch.where(x <= 0.04045, x / 12.92, im)


def lrgb2srgb(im: torch.Tensor) -> torch.Tensor:
    return torch.where(im <= 0.0031308, 12.92 * im, 1.055 * torch.pow(im.clamp_min(0), 1 / 2.4) - 0.055)


def compute_smog(x: torch.Tensor, d: torch.Tensor, airlight=0.76, beta=2.0, vr=1.0, yellow_color=(224, 192, 29),
                 alpha=20.0) -> torch.Tensor:
    """HazeRD smog model with the parameters of shared/trainer/events.yaml:9-14."""
    irradiance = srgb2lrgb(x)
    d = normalize(d, mini=0.3, maxi=1.0)
    d = 1.0 / d
    d = normalize(d, mini=0.1, maxi=1)
    d = F.interpolate(d, size=x.shape[-2:], mode="bilinear", align_corners=True).repeat(1, 3, 1, 1)
    transmission = torch.exp(d * -(beta / vr))
    smogged = lrgb2srgb(transmission * irradiance + (1 - transmission) * airlight)
    a = alpha / 255
    yellow = (torch.tensor(yellow_color, dtype=x.dtype) / 255).view(1, 3, 1, 1)
    return smogged * (1 - a) + yellow * a


# --------------------------------------------------------------------------------------------------
# Cloudy painting: OmniGenerator.paint_cloudy (generator.py:299-328), tutils.rand_perlin_2d / mix_noise (:647-694)
# --------------------------------------------------------------------------------------------------
def perlin_2d(shape, res, angles: torch.Tensor) -> torch.Tensor:
    """``rand_perlin_2d`` with the lattice angles given (the reference draws ``2 pi torch.rand(res+1)``)."""
    import math

    delta = (res[0] / shape[0], res[1] / shape[1])
    d = (shape[0] // res[0], shape[1] // res[1])
    grid = torch.stack(torch.meshgrid(torch.arange(0, res[0], delta[0]), torch.arange(0, res[1], delta[1]),
                                      indexing="ij"), dim=-1) % 1
    grid = grid[: shape[0], : shape[1]]
    gradients = torch.stack((torch.cos(angles), torch.sin(angles)), dim=-1)

    def tile(s1, s2):
        return gradients[s1[0]:s1[1], s2[0]:s2[1]].repeat_interleave(d[0], 0).repeat_interleave(d[1], 1)

    def dot(grad, shift):
        return (torch.stack((grid[..., 0] + shift[0], grid[..., 1] + shift[1]), dim=-1)
                * grad[: shape[0], : shape[1]]).sum(dim=-1)

    n00 = dot(tile([0, -1], [0, -1]), [0, 0])
    n10 = dot(tile([1, None], [0, -1]), [-1, 0])
    n01 = dot(tile([0, -1], [1, None]), [0, -1])
    n11 = dot(tile([1, None], [1, None]), [-1, -1])
    t = 6 * grid ** 5 - 15 * grid ** 4 + 10 * grid ** 3
    return math.sqrt(2) * torch.lerp(torch.lerp(n00, n10, t[..., 0]), torch.lerp(n01, n11, t[..., 0]), t[..., 1])


def paint_cloudy(sd_p: SD, m, x, s, z_h, z_w, angles, sky_idx=9, res=(8, 8), weight=0.8, update=True):
    """``OmniGenerator.paint_cloudy``: sd_p = painter state dict, s = segmentation logits [B,11,h,w]."""
    sky = (torch.argmax(F.interpolate(s, x.shape[-2:], mode="bilinear"), dim=1, keepdim=True) == sky_idx).to(x.dtype)
    noise = perlin_2d(tuple(x.shape[-2:]), res, angles)[None, None]
    noise = noise - noise.min()
    mask = sky.repeat(1, 3, 1, 1)
    noised = mask * (weight * noise + (1 - weight) * x) + (1 - mask) * x
    fake = paint(sd_p, m, noised, z_h, z_w, no_paste=True, update=update)
    return x * (1.0 - m) + fake * m


# --------------------------------------------------------------------------------------------------
# SPADE mask decoder: MaskSpadeDecoder (masker.py:59-231), OmniGenerator.make_m_cond (generator.py:196-230)
# --------------------------------------------------------------------------------------------------
def make_m_cond(d, s, x=None):
    cats = [normalize(d), torch.softmax(s, dim=1)]
    if x is not None:
        cats.append(F.interpolate(x, s.shape[-2:], mode="bilinear", align_corners=True))
    return torch.cat(cats, dim=1)


def mask_spade_decoder(z, cond, sd: SD, prefix: str, num_layers=3, update=True):
    """``MaskSpadeDecoder.forward`` with use_proj (defaults.yaml:174-175), eval-mode BatchNorm everywhere."""
    p = prefix + "." if prefix else ""
    z_h, z_l = z
    z_l = conv2d_block(z_l, sd, p + "low_level_conv", 3, 1, "reflect", "spectral_batch", "lrelu", update)
    z_l = F.interpolate(z_l, size=z_h.shape[-2:], mode="bilinear")
    z_h = conv2d_block(z_h, sd, p + "high_level_conv", 3, 1, "reflect", "spectral_batch", "lrelu", update)
    y = conv2d_block(torch.cat([z_h, z_l], 1), sd, p + "merge_feats_conv", 3, 1, "reflect", "spectral_batch", "lrelu",
                     update)
    for i in range(num_layers):
        y = spade_resnet_block(y, cond, sd, "%sspade_blocks.%d" % (p, i), norm_type="batch", last_activation="lrelu",
                               training=False, update=update)
        y = nearest_resize(y, (y.shape[2] * 2, y.shape[3] * 2))
    return conv2d_block(y, sd, p + "mask_conv", 3, 1, "reflect", "spectral", "none", update)


# --------------------------------------------------------------------------------------------------
# Wildfire event: fire.add_fire (climategan/fire.py:68-126).  PARITY UNPINNED: the torchvision (==0.8.x)
# adjust_contrast / adjust_brightness and kornia (==0.5.10) get_gaussian_kernel2d / filter2d calls are restated from
# those libraries' documented formulas -- neither package is available here, so nothing below was checked against them.
# --------------------------------------------------------------------------------------------------
def _tv_adjust_brightness_u8(img_u8: torch.Tensor, f: float) -> torch.Tensor:
    return (f * img_u8.float()).clamp(0, 255).to(torch.uint8)


def _tv_adjust_contrast_u8(img_u8: torch.Tensor, f: float) -> torch.Tensor:
    r, g, b = img_u8[:, 0].float(), img_u8[:, 1].float(), img_u8[:, 2].float()
    gray = (0.2989 * r + 0.587 * g + 0.114 * b).to(torch.uint8).float()
    mean = gray.mean(dim=(-2, -1), keepdim=True).unsqueeze(1)
    return (f * img_u8.float() + (1 - f) * mean).clamp(0, 255).to(torch.uint8)


def _kornia_gaussian_1d(ks: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    g = torch.exp(-x.pow(2) / (2 * sigma ** 2))
    return g / g.sum()


def add_fire(x, seg_preds, filter_green: float, kernel_size=281, kernel_sigma=140.5, crop_bottom=True, sky_idx=9):
    """``add_fire``: x [B,3,H,W] in [-1,1], seg_preds [B,C,h,w] logits -> float image in [0, 255]."""
    w_t = normalize(x, 0, 255).clone()
    w_t[:, 2] -= 20
    w_t[:, 1] -= 10
    w_t[:, 0] += 40
    w_t = w_t.clamp(0, 255).to(torch.uint8)
    w_t = _tv_adjust_contrast_u8(w_t, 1.5)
    w_t = _tv_adjust_brightness_u8(w_t, 0.73)
    sky = (torch.argmax(seg_preds, dim=1) == sky_idx).unsqueeze(1)
    if crop_bottom:
        i = 2 * sky.shape[-2] // 3
        sky = sky.clone()
        sky[..., i:, :] = 0
    sky = F.interpolate(sky.float(), (w_t.shape[-2], w_t.shape[-1]))
    n_lines, n_cols = int(0.18 * sky.shape[-2]), int(0.18 * sky.shape[-1])          # increase_sky_mask(mask, .18, .18)
    if n_cols > 1:
        sky = F.max_pool2d(sky, (1, 2 * n_cols - 1), stride=1, padding=(0, n_cols - 1))
    if n_lines > 1:
        sky = F.max_pool2d(sky, (2 * n_lines - 1, 1), stride=1, padding=(n_lines - 1, 0))
    g = _kornia_gaussian_1d(kernel_size, kernel_sigma)
    half = kernel_size // 2
    sky = F.conv2d(F.pad(sky, (half, half, 0, 0), mode="reflect"), g.view(1, 1, 1, -1))
    sky = F.conv2d(F.pad(sky, (0, 0, half, half), mode="reflect"), g.view(1, 1, -1, 1))
    filt = torch.ones(w_t.shape)
    filt[:, 0] = 255
    filt[:, 1] = filter_green
    filt[:, 2] = 0
    mk = 200 / 255.0 * sky
    w_t = mk * filt + (1.0 - mk) * w_t
    w_t = _tv_adjust_brightness_u8(w_t.to(torch.uint8), 0.8).float()
    w_t[:, :, 0, 0] = 255.0
    w_t[:, :, -1, -1] = 0.0
    return w_t


# ------------------------------------------------------------------------------------------------ input pipeline (N3)
def skimage_resize_018(image, output_shape):
    """RESTATEMENT of scikit-image 0.18.3 ``skimage.transform.resize(image, output_shape, preserve_range=True,
    anti_aliasing=True)`` for an [H, W, C] image and a (rows, cols) shape (order 1, mode 'reflect', clip True) --
    **parity unpinned**: scikit-image (requirements-3.8.2.txt:69) is a third-party dependency that is neither in
    /root/reference nor installed here, so this follows the library's published source (transform/_warps.py: resize,
    warp; _shared/interpolation.pxd: bilinear_interpolation, coord_map) and is anchored only on the reference's call site
    apply_events.py:230.  The Gaussian pre-filter IS the library's own call (scipy.ndimage.gaussian_filter, installed).
    The one deliberate deviation: skimage estimates the scale / offset of the warp with a least-squares AffineTransform
    (last-ulp jitter on the sampling coordinates); the closed form is used here."""
    import numpy as np
    from scipy import ndimage as ndi

    image = np.asarray(image).astype(np.float64)                        # convert_to_float(image, preserve_range=True)
    rows, cols = output_shape
    h, w = image.shape[:2]
    factors = np.array([h / rows, w / cols, 1.0])
    sigma = np.maximum(0, (factors - 1) / 2)
    image = ndi.gaussian_filter(image, sigma, cval=0, mode="mirror")    # _to_ndimage_mode("reflect") == "mirror"
    r = factors[0] * (np.arange(rows) + 0.5) - 0.5                      # pixel 0 sits at (0.5, 0.5)
    c = factors[1] * (np.arange(cols) + 0.5) - 0.5

    def mirror(i, n):                                                   # coord_map mode 'R'
        i = np.where(i < 0, -i, i)
        return np.where(i >= n, 2 * (n - 1) - i, i)

    minr, maxr = np.floor(r).astype(int), np.ceil(r).astype(int)
    minc, maxc = np.floor(c).astype(int), np.ceil(c).astype(int)
    dr, dc = (r - minr)[:, None, None], (c - minc)[None, :, None]
    r0, r1, c0, c1 = mirror(minr, h), mirror(maxr, h), mirror(minc, w), mirror(maxc, w)
    top = (1 - dc) * image[r0][:, c0] + dc * image[r0][:, c1]
    bottom = (1 - dc) * image[r1][:, c0] + dc * image[r1][:, c1]
    out = (1 - dr) * top + dr * bottom
    return np.clip(out, image.min(), image.max())                       # clip=True


def resize_and_crop(img, to=640):
    """apply_events.py:211-241: aspect-preserving resize (smaller side = ``to``), uint8 truncation, centre crop, / 255."""
    import numpy as np

    h, w = img.shape[:2]
    size = (to, int(to * w / h)) if h < w else (int(to * h / w), to)
    r_img = skimage_resize_018(img, size).astype(np.uint8)
    H, W = r_img.shape[:2]
    top, left = (H - to) // 2, (W - to) // 2
    return r_img[top:top + to, left:left + to, :] / 255.0


def to_m1_p1(img):
    """apply_events.py:179-195"""
    import numpy as np

    if img.min() >= 0 and img.max() <= 1:
        return (img.astype(np.float32) - 0.5) * 2
    raise ValueError("Data range mismatch for image : ({}, {})".format(img.min(), img.max()))


def resize_keep_ratio(img, max_im_width=-1):
    """keep_ratio branch of apply_events.py:494-497 + 502: ``to_m1_p1(resize(img, to_128(img, w), anti_aliasing=True))``.
    Without ``preserve_range`` skimage first scales uint8 to [0, 1] (img_as_float); the filter and the warp are linear, so
    the restatement divides afterwards (differences: last-bit rounding of the float64 pipeline)."""
    import numpy as np

    h, w = img.shape[:2]
    w_target = w if max_im_width < 0 else max_im_width
    nw = int(w_target / 128) * 128
    nh = int(nw * (h / w) / 128) * 128
    return to_m1_p1(np.clip(skimage_resize_018(img, (nh, nw)) / 255.0, 0.0, 1.0))


# ------------------------------------------------------------------------------------------------ masker losses (A19)
def cross_entropy(logits, target):
    """losses.py:106-112: nn.CrossEntropyLoss() on [N, C, H, W] logits and [N, H, W] class ids (mean over pixels)."""
    lp = torch.log_softmax(logits, dim=1)
    return -lp.gather(1, target.long().unsqueeze(1)).mean()


def tv_loss(x, weight=1.0):
    """losses.py:142-169: 2 w (sum dh^2 / count_h + sum dw^2 / count_w) / batch, counts = C (H-1) W and C H (W-1)."""
    b, c, h, w = x.shape
    h_tv = ((x[:, :, 1:, :] - x[:, :, :-1, :]) ** 2).sum()
    w_tv = ((x[:, :, :, 1:] - x[:, :, :, :-1]) ** 2).sum()
    return weight * 2 * (h_tv / (c * (h - 1) * w) + w_tv / (c * h * (w - 1))) / b


def prob_2_entropy(prob):
    """losses.py:453-458: -p log2(p + 1e-30) / log2(C)."""
    import numpy as np

    return -prob * torch.log2(prob + 1e-30) / np.log2(prob.shape[1])


def minent_loss(pred, version=1, lambda_var=0.1):
    """losses.py:172-196: mean entropy per pixel (v1); v2 adds lambda_var times the squared deviation from that mean."""
    n, c, h, w = pred.shape
    ent = prob_2_entropy(pred)
    if version == 1:
        return ent.sum() / (n * h * w)
    dem = ent - ent.sum() / (n * h * w)
    return (ent + lambda_var * dem * dem).sum() / (n * h * w)


def ground_intersection_loss(pred, pseudo_ground):
    """losses.py:444-450: mean of 1[(ground - pred) > 0.5]."""
    return ((pseudo_ground - pred) > 0.5).float().mean()


def advent_wgan(d_out, target):
    """The WGAN branch of ADVENTAdversarialLoss (losses.py:498-499): -mean(y D + (1 - y)(1 - D))."""
    return -(target * d_out + (1 - target) * (1 - d_out)).mean()


def sigm_loss(prediction, target, gmweight=0.5, scale=4):
    """losses.py:237-278 (MiDaS scale-and-shift-invariant loss): medians (torch.median: the LOWER middle element), mean
    absolute deviations, residual R; 4-scale Sobel gradient-matching term.  Quirk reproduced: the Sobel filters are
    expanded to [B, 1, 3, 3] and applied by F.conv2d to the 1-channel residual, so every image's maps are computed with
    B identical output channels and the summed term carries a factor B."""
    t_pred, t_targ = torch.median(prediction), torch.median(target)
    s_pred, s_targ = (prediction - t_pred).abs().mean(), (target - t_targ).abs().mean()
    R = (prediction - t_pred) / s_pred - (target - t_targ) / s_targ
    b = prediction.shape[0]
    num_pix = prediction.shape[-1] * prediction.shape[-2]
    sobelx = torch.tensor([[1.0, 0, -1], [2, 0, -2], [1, 0, -1]]).expand(b, 1, 3, 3)
    sobely = torch.tensor([[1.0, 2, 1], [0, 0, 0], [-1, -2, -1]]).expand(b, 1, 3, 3)
    gm = 0
    for k in range(scale):
        R_ = F.interpolate(R, scale_factor=1 / 2 ** k)
        gm = gm + F.conv2d(R_, sobelx).abs().sum() + F.conv2d(R_, sobely).abs().sum()
    return 0.5 / num_pix * R.abs().sum() + gmweight / num_pix * gm


# --------------------------------------------------------------------------------------------------
# One training iteration of the default task set [d, s, m, p]: Trainer.update_G + Trainer.update_D
# (trainer.py:989-1032) = get_G_loss (get_masker_loss :1184-1254 with masker_{d,s,m}_loss :1389-1616, get_painter_loss
# :1256-1387) and get_D_loss (:1034-1160), each followed by backward().  SURVEY Appendix A lists the terms.
# --------------------------------------------------------------------------------------------------
DEFAULT_LAMBDAS = {  # shared/trainer/defaults.yaml:278-311
    "d.main": 1.0, "d.gml": 0.5, "s.crossent": 1.0, "s.minent": 0.001, "s.advent": 0.001, "m.bce": 1.0, "m.tv": 1.0,
    "m.gi": 0.05, "advent.ent_main": 0.5, "advent.ent_var": 0.1, "advent.adv_main": 1.0, "p.vgg": 10.0,
    "p.featmatch": 10.0,
}


def _trainable(sd: SD) -> SD:
    """Clones of a state dict with requires_grad on what the reference trains: everything floating point except the
    spectral-norm u / v vectors (``requires_grad=False`` parameters, norms.py:129-130) and the BatchNorm buffers."""
    out = {}
    for k, v in sd.items():
        leaf = k.rsplit(".", 1)[-1]
        frozen = leaf in ("weight_u", "weight_v", "running_mean", "running_var", "num_batches_tracked")
        out[k] = v.clone().requires_grad_(v.is_floating_point() and not frozen)
    return out


def _masker_preds(g: SD, x, s_size, d_size):
    z = resnet101(x, g, "encoder")
    d_pred, z_depth = dada_depth_decoder(z, g, "decoders.d", d_size)
    s_pred = deeplab_v3_decoder(z, g, "decoders.s", s_size, z_depth, use_dada=True)
    logits = mask_base_decoder(z, g, "decoders.m")
    return d_pred, s_pred, logits


def masker_g_loss(g: SD, dd: SD, batch: dict, lam: dict, terms: dict):
    """``get_masker_loss`` for the domains of ``batch`` other than rf, in their order (trainer.py:1200-1254)."""
    total = 0
    for dom, b in batch.items():
        if dom == "rf":
            continue
        x = b["x"]
        d_pred, s_pred, logits = _masker_preds(g, x, tuple(b["s"].shape[-2:]), b["d"].shape[-1])
        l = sigm_loss(d_pred, b["d"], lam["d.gml"]) * lam["d.main"]          # computed, then dropped for domain r
        terms["G.task.d." + dom] = l.detach() if dom == "s" else torch.zeros(())     # (trainer.py:1403-1405)
        if dom == "s":
            total = total + l
            l = cross_entropy(s_pred, b["s"].squeeze(1)) * lam["s.crossent"]
            terms["G.task.s.crossent.s"] = l.detach(); total = total + l
        else:
            sm = torch.softmax(s_pred, dim=1)
            l = minent_loss(sm) * lam["s.minent"]
            terms["G.task.s.minent.r"] = l.detach(); total = total + l
            ent = prob_2_entropy(sm) * d_pred.detach()                        # gen.s.use_dada (trainer.py:1455-1456)
            l = advent_wgan(fc_discriminator(ent, dd, "s.Advent"), 0) * lam["s.advent"]
            terms["G.task.s.advent.r"] = l.detach(); total = total + l
        p = torch.sigmoid(logits)
        prob = torch.cat([p, 1 - p], dim=1)
        l = tv_loss(p) * lam["m.tv"]
        terms["G.task.m.tv." + dom] = l.detach(); total = total + l
        if dom == "s":
            l = F.binary_cross_entropy_with_logits(logits, b["m"]) * lam["m.bce"]
            terms["G.task.m.bce.s"] = l.detach(); total = total + l
        else:
            l = ground_intersection_loss(p, b["m"]) * lam["m.gi"]
            terms["G.task.m.gi.r"] = l.detach(); total = total + l
            l = minent_loss(prob, 2, lam["advent.ent_var"]) * lam["advent.ent_main"]
            terms["G.task.m.minent.r"] = l.detach(); total = total + l
            l = advent_wgan(fc_discriminator(prob_2_entropy(prob), dd, "m.Advent"), 0) * lam["advent.adv_main"]
            terms["G.task.m.advent.r"] = l.detach(); total = total + l
    return total


class sub_view(dict):
    """A prefix view of a state dict that WRITES THROUGH (the functional modules store the power-iterated u / v back
    into the dict they are given; ``sub()`` would hand them a copy)."""

    def __init__(self, parent: SD, prefix: str):
        super().__init__()
        self._parent, self._p = parent, prefix + "."
        for k, v in parent.items():
            if k.startswith(self._p):
                dict.__setitem__(self, k[len(self._p):], v)

    def __setitem__(self, k, v):
        dict.__setitem__(self, k, v)
        self._parent[self._p + k] = v


def joint_d_loss(g: SD, dd: SD, batch: dict, z_hw, num_D, n_layers, lam: dict, terms: dict):
    """``get_D_loss`` (trainer.py:1034-1160): Painter branch on a fake painted without a graph, ADVENT branches with the
    BCE form (losses.py:440,496-497) on the detached predictions; ``adv_main`` is applied twice, as in the reference
    (masker_s_loss / masker_m_loss and again at trainer.py:1122,1145)."""
    total = 0
    adv = lam["advent.adv_main"]
    acc = {"s": 0, "m": 0}
    for dom, b in batch.items():
        if dom == "rf":
            x, m = b["x"], b["m"]
            with torch.no_grad():
                fake = paint(sub_view(g, "painter"), m, x, z_hw[0], z_hw[1])
            out = multiscale_discriminator(torch.cat([torch.cat([m, x], 1), torch.cat([m, fake], 1)], 0),
                                           sub_view(dd, "p"), num_D, n_layers)
            real_d = [[t[: t.size(0) // 2] for t in p] for p in out]
            fake_d = [[t[t.size(0) // 2:] for t in p] for p in out]
            l = gan_loss(fake_d, False) + gan_loss(real_d, True)
            terms["D.p.gan"] = l.detach(); total = total + l
            continue
        label = {"s": 0.0, "r": 1.0}[dom]
        with torch.no_grad():
            d_pred, s_pred, logits = _masker_preds(g, b["x"], tuple(b["s"].shape[-2:]), b["d"].shape[-1])
            ent_s = prob_2_entropy(torch.softmax(s_pred, dim=1)) * d_pred
            p = torch.sigmoid(logits)
            ent_m = prob_2_entropy(torch.cat([p, 1 - p], dim=1))
        for task, ent in (("s", ent_s), ("m", ent_m)):
            o = fc_discriminator(ent, dd, task + ".Advent")
            l = F.binary_cross_entropy_with_logits(o, torch.full_like(o, label)) * adv * adv
            acc[task] = acc[task] + l.detach(); total = total + l
    for task in ("s", "m"):
        if torch.is_tensor(acc[task]):
            terms["D.%s.Advent" % task] = acc[task]
    return total


def joint_train_step(sd_g: SD, sd_d: SD, sd_vgg: Optional[SD], batch: dict, n_up: int, num_D: int = 3, n_layers: int = 4,
                     lam: Optional[dict] = None, want_grads: bool = True, g_lr: float = 5e-5, g_betas=(0.9, 0.999)):
    """``Trainer.update_G`` (loss, ``backward()``, ExtraAdam extrapolation of G as at ``global_step`` 0) then
    ``Trainer.update_D`` up to and including its ``backward()`` (the D optimizer step changes nothing observable here) on a
    multi-domain batch ``{"r": {x, d, s, m}, "s": {...}, "rf": {x, m}}`` (plain tensors), generator and discriminators in
    training mode.  Returns {"terms": {...}, "g_grads": {...}, "d_grads": {...}, "g_state": sd, "d_state": sd}: loss terms
    under the names the reference logs them (logger.losses.gen / .disc), gradients of every trainable tensor, and the
    states after the step (spectral-norm vectors power-iterated by every forward, BatchNorm running statistics)."""
    lam = dict(DEFAULT_LAMBDAS, **(lam or {}))
    terms = {}
    g, dd = _trainable(sd_g), _trainable(sd_d)
    for v in dd.values():                       # D frozen during the G update (trainer.py:959-962)
        v.requires_grad_(False)
    rf = batch.get("rf")
    z_hw = (rf["x"].shape[-2] // 2 ** n_up, rf["x"].shape[-1] // 2 ** n_up) if rf is not None else None
    with bn_training():
        g_loss = masker_g_loss(g, dd, batch, lam, terms) if any(d != "rf" for d in batch) else 0
        if rf is not None:
            gp = sub_view(g, "painter")
            g_loss = g_loss + painter_g_loss(gp, dd, sd_vgg, rf, z_hw, num_D, n_layers, lam, terms)
        g_keys = [k for k, v in g.items() if v.requires_grad]
        g_grads = {}
        if want_grads:
            grads = torch.autograd.grad(g_loss, [g[k] for k in g_keys], allow_unused=True)
            g_grads = {k: gr for k, gr in zip(g_keys, grads) if gr is not None}
        terms["G.total_loss"] = g_loss.detach()
        g = {k: v.detach() for k, v in g.items()}
        bn_after_g = {k: v.clone() for k, v in g.items() if k.endswith(("running_mean", "running_var"))}
        if g_lr and want_grads:
            # g_opt_step() at global_step 0 (trainer.py:674-683): ExtraAdam extrapolation -- the D update sees the
            # EXTRAPOLATED generator (lr 5e-5, betas (0.9, 0.999), defaults.yaml:73-77)
            keys = list(g_grads)
            opt = ExtraAdamRef([g[k] for k in keys], lr=g_lr, betas=g_betas)
            opt.extrapolation([g_grads[k] for k in keys])
            for k, v in zip(keys, opt.params):
                g[k] = v
        for k, v in dd.items():                 # D trains again (trainer.py:971-973)
            leaf = k.rsplit(".", 1)[-1]
            dd[k] = v.detach().requires_grad_(leaf not in ("weight_u", "weight_v"))
        d_loss = joint_d_loss(g, dd, batch, z_hw, num_D, n_layers, lam, terms)
        d_keys = [k for k, v in dd.items() if v.requires_grad]
        d_grads = {}
        if want_grads:
            grads = torch.autograd.grad(d_loss, [dd[k] for k in d_keys], allow_unused=True)
            d_grads = {k: gr for k, gr in zip(d_keys, grads) if gr is not None}
        terms["D.total_loss"] = d_loss.detach()
    return {"terms": terms, "g_grads": g_grads, "d_grads": d_grads, "g_state": g, "bn_after_update_G": bn_after_g,
            "d_state": {k: v.detach() for k, v in dd.items()}}


def painter_g_loss(gp: SD, dd: SD, vgg: Optional[SD], b: dict, z_hw, num_D, n_layers, lam: dict, terms: dict):
    """``get_painter_loss`` (trainer.py:1256-1387), single-discriminator branch, default lambdas (TV / context /
    reconstruction 0): VGG on ``fake_flooded * m`` with the already pasted image, unscaled GAN term (trainer.py:1369-1371),
    feature matching.  ``gp``: the painter's tensors (a write-through view of the generator's)."""
    x, m = b["x"], b["m"]
    fake = paint(gp, m, x, z_hw[0], z_hw[1])
    total = 0
    if vgg is not None and lam["p.vgg"] != 0:
        l = vgg_loss(vgg, vgg_preprocess(fake * m), vgg_preprocess(x * m)) * lam["p.vgg"]
        terms["G.p.vgg"] = l.detach(); total = total + l
    out = multiscale_discriminator(torch.cat([torch.cat([m, x], 1), torch.cat([m, fake], 1)], 0), sub_view(dd, "p"),
                                   num_D, n_layers)
    real_d = [[t[: t.size(0) // 2] for t in p] for p in out]
    fake_d = [[t[t.size(0) // 2:] for t in p] for p in out]
    l = gan_loss(fake_d, True)
    terms["G.p.gan"] = l.detach(); total = total + l
    l = feat_match_loss(real_d, fake_d) * lam["p.featmatch"]
    terms["G.p.featmatch"] = l.detach(); total = total + l
    return total
