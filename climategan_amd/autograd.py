"""torch.autograd plumbing of the training path: every Function's forward AND backward is a set of HIP kernels behind
the C ABI (``ops``); torch only records the graph, routes 16-bit NHWC gradient tensors between Functions and
accumulates the fp32 parameter gradients into ``.grad`` (what ``g_loss.backward()`` / ``d_loss.backward()`` do in the
reference, trainer.py:1011,1028).

Activations travel as raw 16-bit NHWC tensors ``[N,H,W,Cs]`` (the ``t`` of an ``ops.NHWC``) because autograd tracks
tensors, with the logical channel count passed alongside.
"""
import ctypes as C

import os

import torch

from . import ops


# Gradient scale for fp16 activation gradients (what torch.cuda.amp.GradScaler does for the reference's ``train.amp``
# mode, trainer.py:1011-1013): every loss Function writes its input gradient multiplied by GRAD_SCALE while returning the
# unscaled loss value; all parameter gradients then come out multiplied by GRAD_SCALE and the trainer divides it out
# before the optimizer step.  1.0 = off (bf16 has the range and does not need it).
GRAD_SCALE = 1.0


def set_grad_scale(s: float):
    global GRAD_SCALE
    GRAD_SCALE = float(s)


BN_GROUPS = 1


class bn_groups:
    """``with bn_groups(G):`` -- every training-mode BatchNorm inside treats its input batch as G equal slices that are
    normalised independently (own batch statistics, running statistics updated once per slice, in order): bit for bit
    what G separate forward calls on the slices compute, in one launch sequence.  The trainer uses it to push the real
    and the simulated domain batch through the Masker TOGETHER (reference trainer.py:1200-1254 calls it per domain)."""

    def __init__(self, groups: int):
        self.groups = int(groups)

    def __enter__(self):
        global BN_GROUPS
        self.prev, BN_GROUPS = BN_GROUPS, self.groups

    def __exit__(self, *a):
        global BN_GROUPS
        BN_GROUPS = self.prev


class SplitBatchFn(torch.autograd.Function):
    """x [G*b, ...] -> G contiguous slices [b, ...] along the batch; the backward concatenates the slices' gradients
    (zeros for a slice nobody used) in ONE pass instead of autograd's G zero-filled full-size tensors and G - 1 adds."""

    @staticmethod
    def forward(ctx, x_t, groups):
        b = x_t.shape[0] // groups
        ctx.shape, ctx.groups = x_t.shape, groups
        return tuple(x_t[i * b:(i + 1) * b] for i in range(groups))

    @staticmethod
    def backward(ctx, *grads):
        b = ctx.shape[0] // ctx.groups
        ref = next(g for g in grads if g is not None)
        parts = [g if g is not None else ref.new_zeros((b,) + tuple(ctx.shape[1:])) for g in grads]
        return torch.cat(parts, dim=0), None


def split_batch(x: "ops.NHWC", groups: int):
    """NHWC map of a concatenated batch -> list of per-group NHWC maps (views in the forward)."""
    if groups == 1:
        return [x]
    if x.t.requires_grad and torch.is_grad_enabled():
        return [ops.NHWC(t, x.c) for t in SplitBatchFn.apply(x.t, groups)]
    b = x.t.shape[0] // groups
    return [ops.NHWC(x.t[i * b:(i + 1) * b], x.c) for i in range(groups)]


class ToNchwFn(torch.autograd.Function):
    """16-bit NHWC map -> fp32 NCHW tensor (``cgan_nhwc_to_nchw``; with ``paste_x / paste_m``: x (1 - m) + y m, reference
    generator.py:295-296).  The backward is the opposite layout kernel on the incoming gradient (times m under a paste):
    what lets ``G.paint`` / ``G.mask`` / ``decoders[t](z)`` / ``D[...](x)`` hand out the reference's NCHW tensors WITH
    their graph, so that the reference's trainer.py (which feeds them to torch ops and to the loss classes) binds
    unchanged."""

    @staticmethod
    def forward(ctx, y_t, c, paste_x, paste_m):
        ctx.cfg = (c, y_t.dtype, y_t.shape[3])
        ctx.save_for_backward(paste_m)
        return ops.nhwc_to_nchw(ops.NHWC(y_t, c), paste_x, paste_m)

    @staticmethod
    def backward(ctx, g):
        c, dt, cs = ctx.cfg
        (paste_m,) = ctx.saved_tensors
        # the forward kernel computes x (1 - mask): hand it 1 - m to get g * m
        keep = (1.0 - paste_m.float()) if paste_m is not None else None
        return ops.nchw_to_nhwc(g, dt, cs=cs, mask=keep).t, None, None, None


class FromNchwFn(torch.autograd.Function):
    """fp32 (or 16-bit) NCHW tensor -> 16-bit NHWC map (``cgan_nchw_to_nhwc``; with ``mask``: x (1 - mask), reference
    generator.py:294); the gradient goes back to ``x`` through the opposite layout kernel."""

    @staticmethod
    def forward(ctx, x, dtype, cs, mask):
        y = ops.nchw_to_nhwc(x, dtype, cs=cs, mask=mask)
        ctx.c, ctx.in_dtype = x.shape[1], x.dtype
        ctx.save_for_backward(mask)
        return y.t

    @staticmethod
    def backward(ctx, dy_t):
        (mask,) = ctx.saved_tensors
        dx = ops.nhwc_to_nchw(ops.NHWC(dy_t.contiguous(), ctx.c))
        if mask is not None:
            dx = dx * (1.0 - mask.float())
        return dx.to(ctx.in_dtype), None, None, None


class FromNchwPairFn(torch.autograd.Function):
    """fp32 NCHW tensor -> NHWC (hi | lo) 16-bit pair map with 2C channels, value = hi + lo (the form the first VGG conv
    takes for its pre-processed input, whose magnitudes of 100-150 would lose +-0.5 in one bf16 value); d/dx = d/d(hi)."""

    @staticmethod
    def forward(ctx, x, dtype):
        x = x.float()
        hi = x.to(dtype).float()
        y = ops.nchw_to_nhwc(torch.cat([hi, x - hi], dim=1), dtype)
        ctx.c = x.shape[1]
        return y.t

    @staticmethod
    def backward(ctx, dy_t):
        return ops.nhwc_to_nchw(ops.NHWC(dy_t.contiguous(), 2 * ctx.c))[:, :ctx.c].contiguous(), None


class ResizeNearestFn(torch.autograd.Function):
    """F.interpolate(mode="nearest") to an arbitrary size (the Painter's latent from the conditioning image,
    painter.py:152) with its adjoint: only reached when the conditioning image itself carries a graph (pl4m)."""

    @staticmethod
    def forward(ctx, x_t, c, size, cs_out):
        ctx.cfg = (c, x_t.shape[1], x_t.shape[2], x_t.shape[3])
        return ops.resize_nearest(ops.NHWC(x_t, c), size, cs_out=cs_out).t

    @staticmethod
    def backward(ctx, dy_t):
        c, h, w, cs = ctx.cfg
        return ops.resize_nearest_bwd(ops.NHWC(dy_t.contiguous(), c), (h, w), cs).t, None, None, None


class ConvFn(torch.autograd.Function):
    """y = act(conv(up?(x), w[/sigma]) + b + up?(res)).  ``weight`` is the fp32 OIHW parameter (``weight_bar`` under
    spectral norm, in which case sigma/u/v of THIS forward's power iteration are given and the weight gradient is mapped
    back through ``w_bar / sigma``, reference norms.py:107-112).  ``cfg`` keys: c_in, stride, pad, dilation, act, slope,
    in_upsample, residual_upsample, c_res."""

    @staticmethod
    def forward(ctx, x_t, weight, bias, res_t, packed, cfg, sn):
        x = ops.NHWC(x_t, cfg["c_in"])
        res = ops.NHWC(res_t, weight.shape[0]) if res_t is not None else None
        if cfg.get("mask_input") and (cfg.get("in_upsample", False) or cfg.get("pair_in", False)):
            raise NotImplementedError("ConvFn: mask_input with a folded upsample / a pair map")
        if cfg.get("want_stats") and res is None and cfg["act"] == ops.ACT_NONE and not cfg.get("in_upsample", False):
            # a BatchNorm follows: its statistics come out of this kernel's epilogue (handed over through cfg: the
            # partials are no tensor of the graph)
            y, cfg["stats_out"] = ops.conv2d_with_stats(x, packed, stride=cfg["stride"], pad=cfg["pad"],
                                                        dilation=cfg["dilation"], pad_mode=cfg.get("pad_mode", ops.PAD_ZERO),
                                                        groups=BN_GROUPS)
        else:
            y = ops.conv2d(x, packed, stride=cfg["stride"], pad=cfg["pad"], dilation=cfg["dilation"], act=cfg["act"],
                           slope=cfg["slope"], residual=res, in_upsample=cfg.get("in_upsample", False),
                           residual_upsample=cfg.get("residual_upsample", False),
                           pad_mode=cfg.get("pad_mode", ops.PAD_ZERO))
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.has_res = res_t is not None
        ctx.premasked = False            # see claim_relu_mask: dy arrives with this conv's ReLU derivative applied
        # (sigma, u, v) as used in this forward: private copies (the batched step hands them over already copied)
        ctx.sn = None if sn is None else (tuple(sn) if cfg.get("sn_owned") else tuple(t.clone() for t in sn))
        ctx.dgrad = None
        if ctx.needs_input_grad[0] and not cfg.get("pair_in", False):
            # the data-gradient operator of this call (w / THIS forward's sigma): packed with all the others of the
            # backward pass in one launch (ops.dgrad_prepack_run, Trainer._backward)
            ctx.dgrad = ops.dgrad_register(weight, ctx.sn[0] if ctx.sn is not None else None, x_t.dtype, cfg["stride"])
        ctx.save_for_backward(x_t, weight, y.t if cfg["act"] != ops.ACT_NONE else None)
        return y.t

    @staticmethod
    def backward(ctx, dy_t):
        cfg = ctx.cfg
        x_t, weight, y_t = ctx.saved_tensors
        c_out = weight.shape[0]
        ups = cfg.get("in_upsample", False)
        dy = ops.NHWC(dy_t.contiguous(), c_out)
        if y_t is not None and not ctx.premasked:
            dy = ops.act_bwd(ops.NHWC(y_t, c_out), dy, cfg["act"], cfg["slope"])
        sigma = ctx.sn[0] if ctx.sn is not None else None
        dx_t = dres_t = None
        pair = cfg.get("pair_in", False)     # x is a (hi | lo) pair map: the conv ran on input-duplicated weights
        w_eff = torch.cat([weight.detach(), weight.detach()], 1) if pair else weight
        if ctx.needs_input_grad[0]:
            h_in, w_in = (x_t.shape[1] * 2, x_t.shape[2] * 2) if ups else (x_t.shape[1], x_t.shape[2])
            # ``mask_input`` (claim_relu_mask): x is a ReLU's output read by nothing else -- its derivative rides in the epilogue
            relu_out = ops.NHWC(x_t, cfg["c_in"]) if cfg.get("mask_input") else None
            dx = ops.conv2d_bwd_data(dy, w_eff, (x_t.shape[0], h_in, w_in), stride=cfg["stride"], pad=cfg["pad"],
                                     dilation=cfg["dilation"], sigma=sigma, pad_mode=cfg.get("pad_mode", ops.PAD_ZERO),
                                     prepacked=ctx.dgrad, relu_out=relu_out)
            dx_t = (ops.sumpool2x2(dx) if ups else dx).t
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres_t = (ops.sumpool2x2(dy) if cfg.get("residual_upsample", False) else dy).t
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            def wgrad():
                dw, db = ops.conv2d_bwd_weight(ops.NHWC(x_t, cfg["c_in"]), dy, tuple(w_eff.shape), stride=cfg["stride"],
                                               pad=cfg["pad"], dilation=cfg["dilation"], want_bias=ctx.has_bias,
                                               in_upsample=ups, pad_mode=cfg.get("pad_mode", ops.PAD_ZERO))
                if pair:
                    dw = dw[:, :weight.shape[1]] + dw[:, weight.shape[1]:]
                if ctx.sn is not None:
                    dw = ops.spectral_norm_bwd(dw, weight.detach(), ctx.sn[1], ctx.sn[2], ctx.sn[0])
                return dw, db
            dw, db = wgrad()
        return dx_t, dw, db, dres_t, None, None, None


class ConvPassFn(torch.autograd.Function):
    """(y, x) = (act(conv(x, w) + b), x): a conv without a residual that also hands its input through.  A
    consumer of the second output (the residual add at the end of a ResNet bottleneck; the L1 term of the VGG loss on a
    tapped feature map, losses.Vgg19) sends its gradient back through
    THIS node, and the backward adds it in the data-gradient kernel's epilogue (``conv2d_bwd_data(add=...)``) -- instead
    of x collecting two gradients that the autograd engine sums with an element-wise pass of its own (33 such adds of
    50-200 MB per train step).  ``cfg["act"]`` (optional): the fused activation of the VGG chain's conv + ReLU pairs."""

    @staticmethod
    def forward(ctx, x_t, weight, bias, packed, cfg):
        x = ops.NHWC(x_t, cfg["c_in"])
        if cfg.get("want_stats") and cfg.get("act", ops.ACT_NONE) != ops.ACT_NONE:
            raise NotImplementedError("ConvPassFn: the statistics epilogue belongs to a conv without an activation")
        if cfg.get("want_stats"):
            y, cfg["stats_out"] = ops.conv2d_with_stats(x, packed, stride=cfg["stride"], pad=cfg["pad"],
                                                        dilation=cfg["dilation"], pad_mode=cfg.get("pad_mode", ops.PAD_ZERO),
                                                        groups=BN_GROUPS)
        else:
            y = ops.conv2d(x, packed, stride=cfg["stride"], pad=cfg["pad"], dilation=cfg["dilation"],
                           act=cfg.get("act", ops.ACT_NONE), slope=cfg.get("slope", 0.0),
                           pad_mode=cfg.get("pad_mode", ops.PAD_ZERO))
        ctx.cfg = cfg
        ctx.has_bias = bias is not None
        ctx.premasked = False            # see claim_relu_mask: dy arrives with this conv's ReLU derivative applied
        ctx.dgrad = ops.dgrad_register(weight, None, x_t.dtype, cfg["stride"]) if ctx.needs_input_grad[0] else None
        ctx.save_for_backward(x_t, weight, y.t if cfg.get("act", ops.ACT_NONE) != ops.ACT_NONE else None)
        return y.t, x_t                   # (returned as-is: autograd makes it an output of this node)

    @staticmethod
    def backward(ctx, dy_t, dpass_t):
        cfg = ctx.cfg
        x_t, weight, y_t = ctx.saved_tensors
        c_out = weight.shape[0]
        dy = ops.NHWC(dy_t.contiguous(), c_out)
        if y_t is not None and not ctx.premasked:
            dy = ops.act_bwd(ops.NHWC(y_t, c_out), dy, cfg["act"], cfg.get("slope", 0.0))
        dx_t = None
        if ctx.needs_input_grad[0]:
            add = ops.NHWC(dpass_t.contiguous(), cfg["c_in"]) if dpass_t is not None else None
            # ``mask_input`` (claim_relu_mask): x is the output of a BatchNormActFn ReLU whose ONLY consumer is this node -- the
            # ReLU's derivative is applied to the summed gradient here, in the data-gradient kernel's epilogue, and that
            # node's backward skips its own mask (one read of `out` and one write of the masked gradient less)
            relu_out = ops.NHWC(x_t, cfg["c_in"]) if cfg.get("mask_input") else None
            dx_t = ops.conv2d_bwd_data(dy, weight, (x_t.shape[0], x_t.shape[1], x_t.shape[2]), stride=cfg["stride"],
                                       pad=cfg["pad"], dilation=cfg["dilation"],
                                       pad_mode=cfg.get("pad_mode", ops.PAD_ZERO), add=add, prepacked=ctx.dgrad,
                                       relu_out=relu_out).t
        dw = db = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.conv2d_bwd_weight(
                ops.NHWC(x_t, cfg["c_in"]), dy, tuple(weight.shape), stride=cfg["stride"], pad=cfg["pad"],
                dilation=cfg["dilation"], want_bias=ctx.has_bias, pad_mode=cfg.get("pad_mode", ops.PAD_ZERO))
        return dx_t, dw, db, None, None


class InstNormActFn(torch.autograd.Function):
    """out = act(instance_norm(x)) (affine-free, biased variance; act none or LeakyReLU)."""

    @staticmethod
    def forward(ctx, x_t, c, eps, act, slope):
        x = ops.NHWC(x_t, c)
        mean, rstd = ops.instnorm_stats(x, eps=eps)
        out = ops.norm_act_apply(x, mean, rstd, act=act, slope=slope)
        ctx.c, ctx.act, ctx.slope = c, act, slope
        ctx.save_for_backward(out.t, rstd)
        return out.t

    @staticmethod
    def backward(ctx, dy_t):
        out_t, rstd = ctx.saved_tensors
        dx = ops.instnorm_act_bwd(ops.NHWC(out_t, ctx.c), ops.NHWC(dy_t.contiguous(), ctx.c), rstd, act=ctx.act,
                                  slope=ctx.slope)
        return dx.t, None, None, None, None


_SPADE_REMAT_GAMMA = os.environ.get("CGAN_SPADE_REMAT_GAMMA") == "1"


_SPADE_FUSED_BWD = os.environ.get("CGAN_SPADE_FUSED_BWD", "1") != "0"     # same-box A/B switch (0: the unfused backward)


class SpadeFn(torch.autograd.Function):
    """y = act(param_free_norm(up?(x)) * (1 + gamma(cond)) + beta(cond)) (reference norms.py:174-186 + the block's
    LeakyReLU); the norm is an instance norm (Painter) or, with cfg["batch_stats"], a training-mode batch norm whose
    (mean, rstd) rows are the batch statistics repeated per sample (MaskSpadeDecoder).  Forward: the fused HIP kernel (the
    128-channel hidden map never leaves LDS; in training it also writes gamma).  Backward: an elementwise stage splits dy into
    the gradients of gamma / beta / the normalised input; the hidden map is re-materialised once for the gamma||beta weight
    gradient; below that, for the Painter's <= 4-channel conditioning image on maps of 80 x 80 and up, ONE fused kernel
    (cgan_spade_hidden_bwd, round 5) turns dgb into mlp_shared's weight / bias gradient -- data gradient of the gamma||beta conv,
    ReLU mask from a re-computed hidden tile, contraction with the conditioning neighbourhood -- without writing the hidden
    gradient; elsewhere (small maps, the SPADE mask decoder's 15-channel conditioning map, a conditioning map that wants a
    gradient itself) the data-gradient conv with the ReLU derivative in its epilogue and a separate weight gradient do the
    same.  The instance-norm backward then gives dx."""

    @staticmethod
    def forward(ctx, x_t, cond_t, mean, rstd, w_sh, b_sh, w_g, b_g, w_b, b_b, packed, cfg):
        x = ops.NHWC(x_t, cfg["c"])
        cond = ops.NHWC(cond_t, cfg["cond_c"])
        # training: the kernel also writes gamma (2 bytes per element) -- the backward then does not re-run mlp_gamma's
        # 128 -> C convolution over the re-materialised hidden map (CGAN_SPADE_REMAT_GAMMA=1: the old path, for A/B)
        gamma_t = None
        if _SPADE_REMAT_GAMMA or not any(ctx.needs_input_grad):
            y = ops.spade_fused(x, mean, rstd, cond, packed, act=cfg["act"], slope=cfg["slope"],
                                x_upsample=cfg["x_upsample"])
        else:
            y, gamma = ops.spade_fused(x, mean, rstd, cond, packed, act=cfg["act"], slope=cfg["slope"],
                                       x_upsample=cfg["x_upsample"], want_gamma=True)
            gamma_t = gamma.t
        ctx.cfg = cfg
        ctx.save_for_backward(x_t, cond_t, mean, rstd, y.t, w_sh, b_sh, w_g, b_g, w_b, b_b, gamma_t)
        return y.t

    @staticmethod
    def backward(ctx, dy_t):
        cfg = ctx.cfg
        x_t, cond_t, mean, rstd, y_t, w_sh, b_sh, w_g, b_g, w_b, b_b, gamma_t = ctx.saved_tensors
        c, dt = cfg["c"], y_t.dtype
        x, y = ops.NHWC(x_t, c), ops.NHWC(y_t, c)
        dy = ops.NHWC(dy_t.contiguous(), c)
        h, w = y.h, y.w
        # re-materialise seg -> hidden -> gamma at full resolution
        seg = ops.resize_nearest(ops.NHWC(cond_t, cfg["cond_c"]), (h, w), cs_out=ops.cs8(cfg["cond_c"]))
        pw_sh = ops.pack_conv_weight(w_sh, b_sh, dt)
        actv = ops.conv2d(seg, pw_sh, pad=1, act=ops.ACT_RELU)
        gamma = ops.NHWC(gamma_t, c) if gamma_t is not None else ops.conv2d(actv, ops.pack_conv_weight(w_g, b_g, dt), pad=1)
        dgb, xhat, dxhat = ops.spade_bwd_prepare(dy, y, x, mean, rstd, gamma, act=cfg["act"], slope=cfg["slope"],
                                                 x_upsample=cfg["x_upsample"])
        del gamma
        # mlp_gamma / mlp_beta as ONE conv with 2C outputs: weight and bias gradients, then the hidden map's gradient
        w_gb = torch.cat([w_g.detach(), w_b.detach()], dim=0)
        want_gb, want_sh = any(ctx.needs_input_grad[6:10]), any(ctx.needs_input_grad[4:6])   # frozen under pl4m
        dw_gb = db_gb = dw_sh = db_sh = None
        if want_gb:
            dw_gb, db_gb = ops.conv2d_bwd_weight(actv, dgb, tuple(w_gb.shape), pad=1)
        # Fused form (round 5, cgan_spade_hidden_bwd): for a <= 4-channel conditioning image that wants no gradient of its
        # own (the Painter) on maps of 80 x 80 and up, mlp_shared's gradient comes out of ONE kernel that re-computes the hidden
        # tile and keeps its gradient on the chip -- no 128-channel gradient map, no separate weight-gradient launch.
        # (the fused kernel is not batch-sliced and addresses dgb with 32-bit byte offsets: larger maps keep the unfused path,
        # whose ops slice the batch -- advisor, round 5)
        fused = (_SPADE_FUSED_BWD and want_sh and not ctx.needs_input_grad[1] and cfg["cond_c"] <= 4 and h * w >= 6400
                 and w_sh.shape[0] == 128 and dgb.t.nbytes < ops.ABI_MAX_BYTES and seg.t.nbytes < ops.ABI_MAX_BYTES)
        d_pre = None
        if fused:
            dw_sh, db_sh = ops.spade_hidden_bwd(dgb, w_gb, seg, pw_sh, c)
        elif want_sh or ctx.needs_input_grad[1]:
            # (the ReLU derivative of mlp_shared rides in the data-gradient kernel's epilogue: no pass over the 128-channel map)
            d_pre = ops.conv2d_bwd_data(dgb, w_gb, (actv.n, h, w), pad=1, relu_out=actv)
        del dgb, actv
        if want_sh and not fused:
            dw_sh, db_sh = ops.conv2d_bwd_weight(seg, d_pre, tuple(w_sh.shape), pad=1)
        dcond_t = None
        if ctx.needs_input_grad[1]:
            # the conditioning map is a prediction (SPADE mask decoder with gen.m.spade.detach = false): back through
            # mlp_shared's conv and the nearest resize
            d_seg = ops.conv2d_bwd_data(d_pre, w_sh.detach(), (seg.n, h, w), pad=1)
            dcond_t = ops.resize_nearest_bwd(d_seg, (cond_t.shape[1], cond_t.shape[2]), cond_t.shape[3]).t
        del d_pre
        # instance norm backward on the normalised tensor at full resolution, then back through the folded upsample
        if cfg.get("batch_stats"):
            # batch param-free norm (MaskSpadeDecoder): the same algebra with the sums taken over n, h, w -- the batch
            # viewed as one image
            nn_, hh, ww, cs = xhat.t.shape
            dx = ops.instnorm_act_bwd(ops.NHWC(xhat.t.view(1, nn_ * hh * ww, 1, cs), c),
                                      ops.NHWC(dxhat.t.view(1, nn_ * hh * ww, 1, cs), c), rstd[0:1].contiguous(),
                                      act=ops.ACT_NONE)
            dx = ops.NHWC(dx.t.view(nn_, hh, ww, cs), c)
        else:
            dx = ops.instnorm_act_bwd(xhat, dxhat, rstd, act=ops.ACT_NONE)
        if cfg["x_upsample"]:
            dx = ops.sumpool2x2(dx)
        dx_t = dx.t if ctx.needs_input_grad[0] else None
        if not want_gb:
            return (dx_t, dcond_t, None, None, dw_sh, db_sh, None, None, None, None, None, None)
        return (dx_t, dcond_t, None, None, dw_sh, db_sh, dw_gb[:c].contiguous(), db_gb[:c].contiguous(),
                dw_gb[c:].contiguous(), db_gb[c:].contiguous(), None, None)


class MakeMCondFn(torch.autograd.Function):
    """cond = cat[normalize(d), softmax(s), bilinear(x)] (OmniGenerator.make_m_cond, generator.py:196-230) with the
    gradient flowing back into d and s (gen.m.spade.detach = false)."""

    @staticmethod
    def forward(ctx, d_t, s_t, x, s_c):
        ctx.s_c, ctx.with_x = s_c, x is not None
        ctx.save_for_backward(d_t, s_t)
        return ops.make_m_cond(ops.NHWC(d_t, 1), ops.NHWC(s_t, s_c), x).t

    @staticmethod
    def backward(ctx, dcond_t):
        d_t, s_t = ctx.saved_tensors
        cond_c = 1 + ctx.s_c + (3 if ctx.with_x else 0)
        dd, ds = ops.make_m_cond_bwd(ops.NHWC(dcond_t.contiguous(), cond_c), ops.NHWC(d_t, 1), ops.NHWC(s_t, ctx.s_c),
                                     ctx.with_x)
        return dd.t, ds.t, None, None


class PainterHeadsFn(torch.autograd.Function):
    """(d_in, vgg_in) = heads(paste(x, m, fake)): see ops.painter_heads; gradient flows to ``fake`` only."""

    @staticmethod
    def forward(ctx, fake_t, x, m, want_d, want_vgg):
        d_in, v_in = ops.painter_heads(ops.NHWC(fake_t, 3), x, m, fake_t.dtype, want_d, want_vgg)
        ctx.save_for_backward(m)
        ctx.want = (want_d, want_vgg)
        empty = fake_t.new_empty(0)
        return (d_in.t if want_d else empty), (v_in.t if want_vgg else empty)

    @staticmethod
    def backward(ctx, dd_t, dv_t):
        (m,) = ctx.saved_tensors
        want_d, want_vgg = ctx.want
        dd = ops.NHWC(dd_t.contiguous(), 4) if want_d and dd_t is not None else None
        dv = ops.NHWC(dv_t.contiguous(), 6) if want_vgg and dv_t is not None else None   # d/d(hi) = d/d(value)
        return ops.painter_heads_bwd(dd, dv, m).t, None, None, None, None


class AvgPool3x3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_t, c):
        ctx.c, ctx.hw = c, (x_t.shape[1], x_t.shape[2])
        return ops.avgpool3x3s2(ops.NHWC(x_t, c)).t

    @staticmethod
    def backward(ctx, dy_t):
        return ops.avgpool3x3s2_bwd(ops.NHWC(dy_t.contiguous(), ctx.c), ctx.hw).t, None


class MaxPool2x2Fn(torch.autograd.Function):
    """``mask_input`` (claim_relu_mask): x is a ReLU's output read by nothing else; the backward also takes that ReLU's derivative."""

    @staticmethod
    def forward(ctx, x_t, c, mask_input=False):
        ctx.c = c
        ctx.mask_input = bool(mask_input)
        ctx.save_for_backward(x_t)
        return ops.maxpool2x2(ops.NHWC(x_t, c)).t

    @staticmethod
    def backward(ctx, dy_t):
        (x_t,) = ctx.saved_tensors
        return ops.maxpool2x2_bwd(ops.NHWC(x_t, ctx.c), ops.NHWC(dy_t.contiguous(), ctx.c), relu_input=ctx.mask_input).t, None, None


_BN_BWD_READS_OUT = os.environ.get("CGAN_BN_BWD_OUT") == "1"     # same-box A/B switch (tools/gpu_ab_env.sh): the old passes


class BatchNormActFn(torch.autograd.Function):
    """out = act(batch_norm(x) [+ residual]) in TRAINING mode (batch statistics over n, h, w; running statistics
    updated in place, momentum / unbiased variance as nn.BatchNorm2d).  gamma / beta may be None (affine=False).
    The residual (the bottleneck's skip connection, resnet101_v3.py:30-50) rides in the apply kernel; its gradient
    dy * act'(out) is a second output of the backward's apply kernel."""

    @staticmethod
    def forward(ctx, x_t, gamma, beta, running_mean, running_var, c, eps, momentum, act, slope, nbt=None, res_t=None,
                conv_stats=None):
        n, h, w, cs = x_t.shape
        G = BN_GROUPS                                                  # see bn_groups
        if n % G:
            raise ValueError("BatchNormActFn: batch %d does not divide into %d groups" % (n, G))
        npix = n * h * w // G
        x_t = x_t.contiguous()
        flat = ops.NHWC(x_t.view(G, npix, 1, cs), c)                   # G "images" of n/G*h*w pixels
        if conv_stats is not None and npix % conv_stats.chunk_pixels == 0:
            # the producing conv's epilogue already reduced x per chunk of pixels: finalize only (one launch, no pass)
            mean, rstd, mean_f, rstd_f = ops.batchnorm_train_stats_from_partials(
                conv_stats, G, npix, c, gamma, beta, running_mean, running_var, nbt, eps, momentum)
        else:
            # batch statistics + (mean', rstd') for the apply kernel + running statistics + step counter: two launches
            mean, rstd, mean_f, rstd_f = ops.batchnorm_train_stats(flat, gamma, beta, running_mean, running_var, nbt, eps,
                                                                   momentum)
        res = None
        if res_t is not None:
            if res_t.shape != x_t.shape:
                raise ValueError("BatchNormActFn: residual %s does not match %s" % (tuple(res_t.shape), tuple(x_t.shape)))
            res = ops.NHWC(res_t.contiguous().view(G, npix, 1, cs), c)
        out = ops.norm_act_apply(flat, mean_f, rstd_f, act=act, slope=slope, residual=res).t.view(n, h, w, cs)
        ctx.cfg = (c, act, slope, G)
        ctx.has_res = res_t is not None
        ctx.premasked = False            # see claim_relu_mask
        # without a fused residual the backward recomputes act'(.) from x with (mean', rstd') -- the apply kernel's own
        # arithmetic -- and never reads `out`: one map less in each of its two passes
        keep_out = res_t is not None or _BN_BWD_READS_OUT
        ctx.save_for_backward(x_t, out if keep_out else None, mean, rstd, gamma,
                              None if keep_out else mean_f, None if keep_out else rstd_f)
        return out

    @staticmethod
    def backward(ctx, dy_t):
        from . import _lib
        lib = _lib.load()
        x_t, out, mean, rstd, gamma, mean_f, rstd_f = ctx.saved_tensors
        c, act, slope, G = ctx.cfg
        if ctx.premasked:
            act = ops.ACT_NONE           # dy arrives as dy * relu'(out): plain BatchNorm backward, the residual's gradient is dy
        n, h, w, cs = x_t.shape
        nbytes = G * lib.cgan_batchnorm_act_bwd_workspace_bytes(c)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x_t.device)
        dx = torch.empty_like(x_t)
        want_res = ctx.has_res and ctx.needs_input_grad[11]
        dy_t = dy_t.contiguous()
        if want_res and act == ops.ACT_NONE:
            dres = dy_t                                       # no activation: the residual's gradient is dy itself
        else:
            dres = torch.empty_like(x_t) if want_res else None
        dg = db = None
        if gamma is not None:
            dg, db = torch.empty((2, c), dtype=torch.float32, device=x_t.device).unbind(0)   # written by the kernel
        _lib.check(lib.cgan_batchnorm_act_bwd_grouped(
            ops._ptr(x_t), ops._ptr(out), ops._ptr(dy_t), ops._ptr(mean), ops._ptr(rstd), ops._ptr(gamma),
            ops._ptr(mean_f), ops._ptr(rstd_f), ops._ptr(dx), ops._ptr(dg), ops._ptr(db), ops._ptr(dres) if dres is not None and dres is not dy_t else None,
            ops._DT[x_t.dtype], n * h * w, c, G, act, slope, ops._ptr(ws), nbytes, ops._stream()),
            "cgan_batchnorm_act_bwd_grouped")
        return dx, dg, db, None, None, None, None, None, None, None, None, dres, None


def claim_relu_mask(out_t: torch.Tensor) -> bool:
    """Called by the ONE consumer of ``out_t`` (a bottleneck's first conv, which also hands the tensor through to its block's
    skip branch: ConvPassFn; the next conv or max-pool of the VGG-19 chain) before the backward pass: if ``out_t`` is the output
    of a training-mode BatchNorm + residual + ReLU node, or of a conv with a fused ReLU, that node will receive its gradient with the ReLU's derivative already applied (the consumer's data-gradient
    kernel takes it in its epilogue, ``cfg["mask_input"]``) and skips its own mask.  Returns whether the claim holds; the
    caller vouches that nothing else reads ``out_t`` in the graph -- a second consumer's gradient would arrive unmasked."""
    fn = out_t.grad_fn
    if fn is None:
        return False
    if isinstance(fn, BatchNormActFn._backward_cls):
        if not fn.has_res or fn.cfg[1] != ops.ACT_RELU:
            return False
    elif isinstance(fn, (ConvFn._backward_cls, ConvPassFn._backward_cls)):   # conv + bias + ReLU in one kernel (VGG-19: losses.Vgg19)
        if fn.cfg.get("act", ops.ACT_NONE) != ops.ACT_RELU:
            return False
    else:
        return False
    fn.premasked = True
    return True


class BceLogitsFn(torch.autograd.Function):
    """weight * sum BCEWithLogits(x, target) over the logical channels -> fp32 device scalar."""

    @staticmethod
    def forward(ctx, x_t, c, target, weight):
        ctx.c = c
        acc = torch.zeros(1, dtype=torch.float32, device=x_t.device)
        dx = ops.bce_logits(ops.NHWC(x_t, c), target, weight * GRAD_SCALE, acc, want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(dx.t if dx is not None else None)
        return acc[0] / GRAD_SCALE if GRAD_SCALE != 1.0 else acc[0]

    @staticmethod
    def backward(ctx, g):
        (dx_t,) = ctx.saved_tensors
        if dx_t is None:
            return None, None, None, None
        return ops.scale_by_scalar(ops.NHWC(dx_t, ctx.c), g.reshape(1).float().contiguous()).t, None, None, None


class MseConstFn(torch.autograd.Function):
    """weight * sum (x - target)^2 over the logical channels (GANLoss with use_lsgan=True, losses.py:50-52)."""

    @staticmethod
    def forward(ctx, x_t, c, target, weight):
        from . import _lib
        ctx.c = c
        acc = torch.zeros(1, dtype=torch.float32, device=x_t.device)
        dx = torch.empty_like(x_t) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.load().cgan_mse_const_nhwc(ops._ptr(x_t), ops._DT[x_t.dtype], _npix(x_t), c, float(target),
                                                   float(weight * GRAD_SCALE), ops._ptr(acc), ops._ptr(dx), ops._stream()),
                   "cgan_mse_const_nhwc")
        ctx.save_for_backward(dx)
        return acc[0] / GRAD_SCALE if GRAD_SCALE != 1.0 else acc[0]

    @staticmethod
    def backward(ctx, g):
        (dx_t,) = ctx.saved_tensors
        if dx_t is None:
            return None, None, None, None
        return ops.scale_by_scalar(ops.NHWC(dx_t, ctx.c), g.reshape(1).float().contiguous()).t, None, None, None


class PainterAuxFn(torch.autograd.Function):
    """The Painter's optional image-space terms on the pasted image (trainer.py:1289-1315): returns (their sum, the three
    weighted values tv / context / reconstruction as a detached [3] tensor); the gradient flows to ``fake`` only."""

    @staticmethod
    def forward(ctx, fake_t, x, m, lam_tv, lam_ctx, lam_rec):
        from . import _lib
        n, h, w = fake_t.shape[0], fake_t.shape[1], fake_t.shape[2]
        x = x.contiguous().float()
        m = m.contiguous().float()
        cnt = float(n * 3 * h * w)
        wh = GRAD_SCALE * lam_tv * 2.0 / (3 * (h - 1) * w) / n                 # TVLoss.forward, losses.py:157-166
        ww = GRAD_SCALE * lam_tv * 2.0 / (3 * h * (w - 1)) / n
        acc = torch.zeros(3, dtype=torch.float32, device=fake_t.device)
        dfake = torch.empty_like(fake_t) if ctx.needs_input_grad[0] else None
        _lib.check(_lib.load().cgan_painter_aux_losses(ops._ptr(fake_t), ops._ptr(x), ops._ptr(m), ops._DT[fake_t.dtype], n, h, w,
                                                       wh, ww, GRAD_SCALE * lam_ctx / cnt, GRAD_SCALE * lam_rec / cnt,
                                                       ops._ptr(acc), ops._ptr(dfake), ops._stream()), "cgan_painter_aux_losses")
        ctx.save_for_backward(dfake)
        if GRAD_SCALE != 1.0:
            acc = acc / GRAD_SCALE
        ctx.mark_non_differentiable(acc)
        return acc.sum(), acc

    @staticmethod
    def backward(ctx, g, _g_parts):
        (dfake,) = ctx.saved_tensors
        if dfake is None:
            return None, None, None, None, None, None
        return ops.scale_by_scalar(ops.NHWC(dfake, 3), g.reshape(1).float().contiguous()).t, None, None, None, None, None


class ResizeBicubicFn(torch.autograd.Function):
    """F.interpolate(mode="bicubic", align_corners=False) with its adjoint (the DADA depth decoder's resize when the map's
    width differs from the target size, depth.py:143-149)."""

    @staticmethod
    def forward(ctx, x_t, c, size):
        ctx.cfg = (c, x_t.shape[1], x_t.shape[2], int(size[0]), int(size[1]))
        return ops.resize_bicubic(ops.NHWC(x_t, c), size).t

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        c, h_in, w_in, h_out, w_out = ctx.cfg
        dy = dy.contiguous()
        dx = torch.empty((dy.shape[0], h_in, w_in, dy.shape[3]), dtype=dy.dtype, device=dy.device)
        _lib.check(_lib.load().cgan_resize_bicubic_bwd_nhwc(ops._ptr(dy), ops._ptr(dx), ops._DT[dy.dtype], dy.shape[0], c, h_in,
                                                            w_in, h_out, w_out, ops._stream()), "cgan_resize_bicubic_bwd_nhwc")
        return dx, None, None


class HingeFn(torch.autograd.Function):
    """weight * sum of HingeLoss.loss's per-element terms (reference losses.py:565-579) -> fp32 device scalar."""

    @staticmethod
    def forward(ctx, x_t, c, target_is_real, for_discriminator, weight):
        ctx.c = c
        acc = torch.zeros(1, dtype=torch.float32, device=x_t.device)
        dx = ops.hinge_loss(ops.NHWC(x_t, c), target_is_real, for_discriminator, weight * GRAD_SCALE, acc,
                            want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(dx.t if dx is not None else None)
        return acc[0] / GRAD_SCALE if GRAD_SCALE != 1.0 else acc[0]

    @staticmethod
    def backward(ctx, g):
        (dx_t,) = ctx.saved_tensors
        if dx_t is None:
            return None, None, None, None, None
        return ops.scale_by_scalar(ops.NHWC(dx_t, ctx.c), g.reshape(1).float().contiguous()).t, None, None, None, None


class L1Fn(torch.autograd.Function):
    """weight * sum |a - b| (b is a constant, as FeatMatchLoss detaches the real features, losses.py:99-101)."""

    @staticmethod
    def forward(ctx, a_t, b_t, c, weight):
        ctx.c = c
        acc = torch.zeros(1, dtype=torch.float32, device=a_t.device)
        da = ops.l1_loss(ops.NHWC(a_t, c), ops.NHWC(b_t, c), weight * GRAD_SCALE, acc, want_grad=ctx.needs_input_grad[0])
        ctx.save_for_backward(da.t if da is not None else None)
        return acc[0] / GRAD_SCALE if GRAD_SCALE != 1.0 else acc[0]

    @staticmethod
    def backward(ctx, g):
        (da_t,) = ctx.saved_tensors
        if da_t is None:
            return None, None, None, None
        return ops.scale_by_scalar(ops.NHWC(da_t, ctx.c), g.reshape(1).float().contiguous()).t, None, None, None


# ---------------------------------------------------------------------------------------------------------------------
# Masker-side losses and the probability maps they consume
# ---------------------------------------------------------------------------------------------------------------------
def _call(name, *args):
    from . import _lib
    _lib.check(getattr(_lib.load(), name)(*args), name)


def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


class SoftmaxFn(torch.autograd.Function):
    """torch.softmax(s, dim=1) on NHWC logits."""

    @staticmethod
    def forward(ctx, x_t, c):
        y = torch.empty_like(x_t)
        _call("cgan_softmax_nhwc", ops._ptr(x_t), ops._ptr(y), ops._DT[x_t.dtype], _npix(x_t), c, ops._stream())
        ctx.c = c
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        _call("cgan_softmax_bwd_nhwc", ops._ptr(y), ops._ptr(dy.contiguous()), ops._ptr(dx), ops._DT[y.dtype], _npix(y),
              ctx.c, ops._stream())
        return dx, None


class SigmoidPairFn(torch.autograd.Function):
    """cat[sigmoid(x), 1 - sigmoid(x)] of 1-channel NHWC logits -> 2-channel NHWC probabilities."""

    @staticmethod
    def forward(ctx, x_t):
        y = torch.empty_like(x_t)
        _call("cgan_sigmoid_pair_nhwc", ops._ptr(x_t), ops._ptr(y), ops._DT[x_t.dtype], _npix(x_t), ops._stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dx = torch.empty_like(y)
        _call("cgan_sigmoid_pair_bwd_nhwc", ops._ptr(y), ops._ptr(dy.contiguous()), ops._ptr(dx), ops._DT[y.dtype],
              _npix(y), ops._stream())
        return dx


class SigmoidFn(torch.autograd.Function):
    """Elementwise sigmoid of an NHWC map (pad channels come out as 0.5; consumers read logical channels only)."""

    @staticmethod
    def forward(ctx, x_t, c):
        y = ops.sigmoid(ops.NHWC(x_t, c))
        ctx.c = c
        ctx.save_for_backward(y.t)
        return y.t

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ops.act_bwd(ops.NHWC(y, ctx.c), ops.NHWC(dy.contiguous(), ctx.c), ops.ACT_SIGMOID).t, None


class EntropyMapFn(torch.autograd.Function):
    """prob_2_entropy(p) [* depth] (depth: 1-channel NHWC map, a constant)."""

    @staticmethod
    def forward(ctx, p_t, c, depth_t):
        y = torch.empty_like(p_t)
        _call("cgan_entropy_map_nhwc", ops._ptr(p_t), ops._ptr(depth_t), ops._ptr(y), ops._DT[p_t.dtype], _npix(p_t), c,
              ops._stream())
        ctx.c = c
        ctx.save_for_backward(p_t, depth_t)
        return y

    @staticmethod
    def backward(ctx, dy):
        p_t, depth_t = ctx.saved_tensors
        dp = torch.empty_like(p_t)
        _call("cgan_entropy_map_bwd_nhwc", ops._ptr(p_t), ops._ptr(depth_t), ops._ptr(dy.contiguous()), ops._ptr(dp),
              ops._DT[p_t.dtype], _npix(p_t), ctx.c, ops._stream())
        return dp, None, None


class AdventPairFn(torch.autograd.Function):
    """The ADVENT discriminators' input from the logits: prob_2_entropy(softmax(s)) [* depth], or of the mask's
    cat[sigmoid(x), 1 - sigmoid(x)], evaluated in fp32 and stored as a (hi | lo) 16-bit pair with 2C channels
    (``cgan_advent_entropy_pair_nhwc``); the first discriminator conv takes it with duplicated weights (ConvFn
    ``pair_in``).  One 16-bit entropy value is coarser than the signal an untrained prediction carries."""

    @staticmethod
    def forward(ctx, x_t, c, sigmoid_pair, depth_t):
        C = 2 if sigmoid_pair else c
        y = torch.empty(x_t.shape[:-1] + (ops.cs8(2 * C),), dtype=x_t.dtype, device=x_t.device)
        _call("cgan_advent_entropy_pair_nhwc", ops._ptr(x_t), ops._ptr(depth_t), ops._ptr(y), ops._DT[x_t.dtype],
              _npix(x_t), c, int(sigmoid_pair), ops._stream())
        ctx.c, ctx.sig = c, int(sigmoid_pair)
        ctx.save_for_backward(x_t, depth_t)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_t, depth_t = ctx.saved_tensors
        dx = torch.empty_like(x_t)
        _call("cgan_advent_entropy_pair_bwd_nhwc", ops._ptr(x_t), ops._ptr(depth_t), ops._ptr(dy.contiguous()),
              ops._ptr(dx), ops._DT[x_t.dtype], _npix(x_t), ctx.c, ctx.sig, ops._stream())
        return dx, None, None, None


class EntropyPairFromNchwFn(torch.autograd.Function):
    """prob_2_entropy(prob) [* depth] of fp32 NCHW probabilities (the reference's ADVENT call signature, losses.py:517-519)
    -> the (hi | lo) pair map the ADVENT discriminators take; backward: fp32 NCHW d(prob)."""

    @staticmethod
    def forward(ctx, prob, depth, dtype):
        prob = prob.contiguous().float()
        depth = depth.contiguous().float() if depth is not None else None
        n, c, h, w = prob.shape
        if depth is not None and tuple(depth.shape) != (n, 1, h, w):
            raise ValueError("advent: depth %s does not match the probabilities %s" % (tuple(depth.shape), tuple(prob.shape)))
        y = torch.empty((n, h, w, ops.cs8(2 * c)), dtype=dtype, device=prob.device)
        _call("cgan_entropy_pair_from_nchw", ops._ptr(prob), ops._ptr(depth), ops._ptr(y), ops._DT[dtype], n, c, h, w,
              ops._stream())
        ctx.save_for_backward(prob, depth)
        return y

    @staticmethod
    def backward(ctx, dy):
        prob, depth = ctx.saved_tensors
        n, c, h, w = prob.shape
        dp = torch.empty_like(prob)
        _call("cgan_entropy_pair_from_nchw_bwd", ops._ptr(prob), ops._ptr(depth), ops._ptr(dy.contiguous()), ops._ptr(dp),
              ops._DT[dy.dtype], n, c, h, w, ops._stream())
        return dp, None, None


class _ScalarLossFn(torch.autograd.Function):
    """Shared shape of the value+gradient loss kernels: ``run(acc, dx_or_None)`` fills the device scalar and, when the
    input wants a gradient, the gradient of the accumulated term; backward scales it by the upstream scalar."""

    @staticmethod
    def forward(ctx, x_t, c, run):
        acc = torch.zeros(1, dtype=torch.float32, device=x_t.device)
        dx = torch.empty_like(x_t) if ctx.needs_input_grad[0] else None
        run(acc, dx)                      # the closures fold GRAD_SCALE into their weights
        ctx.c = c
        ctx.save_for_backward(dx)
        return acc[0] / GRAD_SCALE if GRAD_SCALE != 1.0 else acc[0]

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        if dx is None:
            return None, None, None
        return ops.scale_by_scalar(ops.NHWC(dx, ctx.c), g.reshape(1).float().contiguous()).t, None, None


def softmax_ce(logits: ops.NHWC, target: torch.Tensor):
    """nn.CrossEntropyLoss (mean over pixels); target int64 [n, h, w]."""
    n = _npix(logits.t)
    tgt = target.contiguous().long()
    return _ScalarLossFn.apply(logits.t, logits.c, lambda acc, dx: _call(
        "cgan_softmax_ce_nhwc", ops._ptr(logits.t), ops._ptr(tgt), logits.dtype_id, n, logits.c, GRAD_SCALE / n, ops._ptr(acc),
        ops._ptr(dx), ops._stream()))


def tv_loss(x: ops.NHWC, tvloss_weight=1.0):
    """TVLoss.forward (losses.py:157-166)."""
    b, h, w, c = x.n, x.h, x.w, x.c
    wh = GRAD_SCALE * tvloss_weight * 2.0 / (c * (h - 1) * w) / b
    ww = GRAD_SCALE * tvloss_weight * 2.0 / (c * h * (w - 1)) / b
    return _ScalarLossFn.apply(x.t, c, lambda acc, dx: _call(
        "cgan_tv_nhwc", ops._ptr(x.t), x.dtype_id, b, h, w, c, wh, ww, ops._ptr(acc), ops._ptr(dx), ops._stream()))


def minent_loss(p: ops.NHWC, version=1, lambda_var=0.1):
    """MinentLoss.__call__ (losses.py:185-196) on a probability map."""
    n = _npix(p.t)
    ws = torch.empty(4096, dtype=torch.float32, device=p.t.device)          # CGAN_MINENT_WORKSPACE_FLOATS
    return _ScalarLossFn.apply(p.t, p.c, lambda acc, dx: _call(
        "cgan_minent_nhwc", ops._ptr(p.t), p.dtype_id, n, p.c, int(version), float(lambda_var), GRAD_SCALE, ops._ptr(acc),
        ops._ptr(dx), ops._ptr(ws), ops._stream()))


def bce_logits_map(x: ops.NHWC, target: torch.Tensor):
    """nn.BCEWithLogitsLoss(x, target) with a target map [n, 1, h, w] (fp32)."""
    n = _npix(x.t)
    tgt = target.contiguous().float()
    return _ScalarLossFn.apply(x.t, x.c, lambda acc, dx: _call(
        "cgan_bce_logits_map_nhwc", ops._ptr(x.t), ops._ptr(tgt), x.dtype_id, n, GRAD_SCALE / n, ops._ptr(acc), ops._ptr(dx),
        ops._stream()))


def ground_intersection(p: ops.NHWC, ground: torch.Tensor):
    """GroundIntersectionLoss (piecewise constant: a detached scalar)."""
    n = _npix(p.t)
    acc = torch.zeros(1, dtype=torch.float32, device=p.t.device)
    g = ground.contiguous().float()
    _call("cgan_ground_intersection_nhwc", ops._ptr(p.t), ops._ptr(g), p.dtype_id, n, 1.0 / n, ops._ptr(acc),
          ops._stream())
    return acc[0]


def advent_wgan(d_out: ops.NHWC, target: float):
    """-mean(y * D + (1 - y) * (1 - D)) (losses.py:498-499) for a scalar domain label y."""
    n = _npix(d_out.t) * d_out.c
    a, b = -GRAD_SCALE * (2.0 * target - 1.0) / n, -GRAD_SCALE * (1.0 - target) / n
    return _ScalarLossFn.apply(d_out.t, d_out.c, lambda acc, dx: _call(
        "cgan_affine_sum_nhwc", ops._ptr(d_out.t), d_out.dtype_id, _npix(d_out.t), d_out.c, a, b, ops._ptr(acc),
        ops._ptr(dx), ops._stream()))


# ---------------------------------------------------------------------------------------------------------------------
# Structural ops of the Masker's graph
# ---------------------------------------------------------------------------------------------------------------------
class ResizeBilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_t, c, size, align_corners):
        y = ops.resize_bilinear(ops.NHWC(x_t, c), size, align_corners=align_corners)
        ctx.cfg = (c, x_t.shape[1], x_t.shape[2], int(size[0]), int(size[1]), bool(align_corners))
        return y.t

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        c, h_in, w_in, h_out, w_out, align = ctx.cfg
        dy = dy.contiguous()
        n = dy.shape[0]
        lib = _lib.load()
        nbytes = lib.cgan_resize_bilinear_bwd_workspace_bytes(n, c, h_in, w_in)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
        dx = torch.empty((n, h_in, w_in, dy.shape[3]), dtype=dy.dtype, device=dy.device)
        _lib.check(lib.cgan_resize_bilinear_bwd_nhwc(ops._ptr(dy), ops._ptr(dx), ops._DT[dy.dtype], n, c, h_in, w_in,
                                                     h_out, w_out, int(align), ops._ptr(ws), nbytes, ops._stream()),
                   "cgan_resize_bilinear_bwd_nhwc")
        return dx, None, None, None


class ResizeNearest2xFn(torch.autograd.Function):
    """InterpolateNearest2d(scale_factor=2) materialised (the mask decoder's upsamples under autograd)."""

    @staticmethod
    def forward(ctx, x_t, c):
        ctx.c = c
        return ops.resize_nearest(ops.NHWC(x_t, c), (x_t.shape[1] * 2, x_t.shape[2] * 2)).t

    @staticmethod
    def backward(ctx, dy):
        return ops.sumpool2x2(ops.NHWC(dy.contiguous(), ctx.c)).t, None


class MaxPool3x3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_t, c):
        ctx.c = c
        ctx.save_for_backward(x_t)
        return ops.maxpool3x3s2(ops.NHWC(x_t, c)).t

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        (x_t,) = ctx.saved_tensors
        dx = torch.empty_like(x_t)
        _lib.check(_lib.load().cgan_maxpool3x3s2_bwd_nhwc(ops._ptr(x_t), ops._ptr(dy.contiguous()), ops._ptr(dx),
                                                          ops._DT[x_t.dtype], x_t.shape[0], ctx.c, x_t.shape[1],
                                                          x_t.shape[2], ops._stream()), "cgan_maxpool3x3s2_bwd_nhwc")
        return dx, None


class AddActFn(torch.autograd.Function):
    """y = act(a + b); the gradient dy * act'(y) goes to both inputs."""

    @staticmethod
    def forward(ctx, a_t, b_t, c, act, slope):
        from . import _lib
        y = torch.empty_like(a_t)
        _lib.check(_lib.load().cgan_add_act_nhwc(ops._ptr(a_t), ops._ptr(b_t), ops._ptr(y), ops._DT[a_t.dtype], act,
                                                 slope, a_t.numel(), ops._stream()), "cgan_add_act_nhwc")
        ctx.cfg = (c, act, slope)
        ctx.save_for_backward(y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        c, act, slope = ctx.cfg
        (y,) = ctx.saved_tensors
        g = dy.contiguous()
        if y is not None:
            g = ops.act_bwd(ops.NHWC(y, c), ops.NHWC(g, c), act, slope).t
        return g, g, None, None, None


class MulFn(torch.autograd.Function):
    """y = a * b elementwise (the DADA fusion z * z_depth, deeplab_v3.py:253-254)."""

    @staticmethod
    def forward(ctx, a_t, b_t, c):
        ctx.c = c
        ctx.save_for_backward(a_t, b_t)
        return ops.eltwise_mul(ops.NHWC(a_t, c), ops.NHWC(b_t, c)).t

    @staticmethod
    def backward(ctx, dy):
        a_t, b_t = ctx.saved_tensors
        g = ops.NHWC(dy.contiguous(), ctx.c)
        da = ops.eltwise_mul(g, ops.NHWC(b_t, ctx.c)).t if ctx.needs_input_grad[0] else None
        db = ops.eltwise_mul(g, ops.NHWC(a_t, ctx.c)).t if ctx.needs_input_grad[1] else None
        return da, db, None


class ConcatFn(torch.autograd.Function):
    """torch.cat along channels of NHWC maps (logical channel counts ``cs``; every offset but the last a multiple of 8)."""

    @staticmethod
    def forward(ctx, cs_list, *ts):
        ctx.cs_list = list(cs_list)
        y = ops.concat_channels([ops.NHWC(t, c) for t, c in zip(ts, cs_list)])
        return y.t

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        lib = _lib.load()
        dy = dy.contiguous()
        npix = dy.shape[0] * dy.shape[1] * dy.shape[2]
        grads, off = [], 0
        for i, c in enumerate(ctx.cs_list):
            if ctx.needs_input_grad[1 + i]:
                g = torch.empty(dy.shape[:3] + (ops.cs8(c),), dtype=dy.dtype, device=dy.device)
                _lib.check(lib.cgan_slice_channels_nhwc(ops._ptr(dy), ops._ptr(g), npix, c, dy.shape[3], off,
                                                        ops._stream()), "cgan_slice_channels_nhwc")
                grads.append(g)
            else:
                grads.append(None)
            off += c
        return (None,) + tuple(grads)


def sigm_loss(pred: ops.NHWC, target: torch.Tensor, gmweight=0.5, scales=4):
    """SIGMLoss.__call__ (losses.py:248-278) on the depth decoder's NHWC map vs a [b, 1, h, w] target."""
    from . import _lib
    lib = _lib.load()
    tgt = target.contiguous().float()
    b, h, w = pred.n, pred.h, pred.w
    nbytes = lib.cgan_sigm_loss_workspace_bytes(b, h, w)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pred.t.device)
    return _ScalarLossFn.apply(pred.t, pred.c, lambda acc, dx: _lib.check(lib.cgan_sigm_loss_nhwc(
        ops._ptr(pred.t), ops._ptr(tgt), pred.dtype_id, b, h, w, float(gmweight), int(scales), GRAD_SCALE, ops._ptr(acc),
        ops._ptr(dx), ops._ptr(ws), nbytes, ops._stream()), "cgan_sigm_loss_nhwc"))
