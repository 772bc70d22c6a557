"""Generate tests/golden/*.npz by running the REAL reference (dev container only; TEST INFRASTRUCTURE).

    python -m oracle.make_golden            # from the repo root, needs /root/reference

Inputs and weights come from the portable fill (climategan_amd/fill.py) so fixtures only hold the
reference's OUTPUTS (and post-forward spectral-norm u/v state).  The committed fixtures are data, not
reference source.  ``golden_cases()`` is shared with the tests so both sides build identical inputs.
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

from climategan_amd import fill

GOLDEN_DIR = Path(__file__).resolve().parent.parent / "tests" / "golden"


# ------------------------------------------------------------------ case table (shared with tests)
def golden_cases():
    return {
        "spade_c20": dict(kind="spade", C=20, cond_nc=3, H=12, W=16, cond_hw=(48, 64), B=2, seed=11),
        "spade_c40": dict(kind="spade", C=40, cond_nc=3, H=8, W=8, cond_hw=(64, 64), B=1, seed=12),
        "resblk_16_8": dict(kind="resblk", fin=16, fout=8, H=10, W=12, cond_hw=(40, 48), B=2, seed=21),
        "resblk_8_8": dict(kind="resblk", fin=8, fout=8, H=10, W=12, cond_hw=(40, 48), B=2, seed=22),
        "painter_up4": dict(kind="painter", latent_dim=32, n_up=4, H=64, W=96, B=2, seed=31, full=True),
        "painter_up7": dict(kind="painter", latent_dim=64, n_up=7, H=256, W=384, B=1, seed=32, full=False),
        "painter_640": dict(kind="painter", latent_dim=640, n_up=7, H=640, W=640, B=1, seed=33, full=False),
        "paint_up4": dict(kind="paint", latent_dim=32, n_up=4, H=64, W=96, B=2, seed=34, full=True),
        "disc_p": dict(kind="disc_p", ndf=8, n_layers=3, num_D=3, H=96, W=128, B=2, seed=41),
        "disc_fc": dict(kind="disc_fc", num_classes=11, H=64, W=96, B=2, seed=42),
        "masker_small": dict(kind="masker", H=128, W=160, B=1, seed=61, gain=1.6),
        "infer_small": dict(kind="infer", H=128, W=160, B=2, seed=71, gain=1.6, latent_dim=32, n_up=4, bin_value=0.43),
        "cloudy_small": dict(kind="cloudy", H=128, W=160, B=2, seed=71, gain=1.6, latent_dim=32, n_up=4, bin_value=0.43,
                             rng_seed=4321, sky_idx=6),   # class 6 covers 14 % of this untrained net's argmax map
        "maskspade_small": dict(kind="maskspade", H=128, W=160, B=2, seed=62, gain=1.6),
        "masker_losses": dict(kind="masker_losses", H=24, W=32, B=2, seed=95),
        "mstep": dict(kind="mstep", H=128, W=160, B=2, seed=66, gain=1.6, sub=512),
        # the same step with the SPADE mask decoder conditioned on the NON-detached predictions (defaults.yaml:168,182)
        "mstep_spade": dict(kind="mstep", H=128, W=160, B=2, seed=67, gain=1.6, sub=512, use_spade=True),
        "dstep_p": dict(kind="dstep_p", ndf=16, n_layers=3, num_D=3, H=96, W=128, B=2, seed=81),
        "gstep_p": dict(kind="gstep_p", latent_dim=32, n_up=4, ndf=16, n_layers=3, num_D=3, H=96, W=128, B=2, seed=91),
        # the non-default branches of get_painter_loss (trainer.py:1289-1315: TV / context / reconstruction, lambdas 0 in
        # defaults.yaml:293-300) and the LSGAN form of GANLoss (losses.py:50-52), on a SOFT mask (0.1 / 0.9: with a binary
        # mask the context term (p - x)(1 - m) = m (1 - m)(fake - x) vanishes identically)
        "gstep_p_aux": dict(kind="gstep_p", latent_dim=32, n_up=4, ndf=16, n_layers=3, num_D=3, H=96, W=128, B=2, seed=92,
                            aux=dict(tv=2.0, context=3.0, reconstruction=5.0, lsgan=True, soft_mask=True)),
        # the local / global discriminator pair (dis.p.use_local_discriminator: trainer.py:1323-1358 on the G side,
        # 1085-1099 on the D side; discriminator.py:242-324 builds the two 3-channel discriminators), lambdas.G.p.gan = 2
        # so that the scaling of this branch (the single-discriminator branch does not scale) shows
        "gstep_p_local": dict(kind="gstep_p", latent_dim=32, n_up=4, ndf=16, n_layers=3, num_D=3, H=96, W=128, B=2, seed=93,
                              local=dict(lambda_gan=2.0)),
        # both non-default branches at once (round 5): the image-space terms of trainer.py:1289-1315 in front of the local /
        # global pair of :1323-1358, on a soft mask (G side only: the D side is gstep_p_local's)
        "gstep_p_local_aux": dict(kind="gstep_p", latent_dim=32, n_up=4, ndf=16, n_layers=3, num_D=3, H=96, W=128, B=2, seed=94,
                                  local=dict(lambda_gan=2.0), aux=dict(tv=2.0, context=3.0, reconstruction=5.0, soft_mask=True)),
        # VGG term of get_painter_loss through the reference's own Vgg19 / VGGLoss / vgg_preprocess; VGG-19 weights from
        # the portable fill with a He-preserving bound (gain sqrt(6): activations keep the input's 0-255 scale)
        "vgg_small": dict(kind="vgg", H=64, W=96, B=2, seed=85, gain=2.449489742783178, lambda_vgg=10.0),
        # fire.add_fire itself (warm / contrast / sky mask / 18 % dilation / blur / paste / brightness); its three
        # third-party calls are bound to the documented formulas (see run_reference_fire)
        "fire_small": dict(kind="fire", H=160, W=192, B=2, seed=86, sky_idx=9, rng_seed=1234),
        "hinge_small": dict(kind="hinge", sizes=[(12, 16), (6, 8), (3, 4)], B=2, seed=87),
        "extra_adam": dict(kind="extra_adam", shapes=[(33, 7), (128,), (5, 3, 3, 3)], steps=4, lr=5e-5, betas=(0.9, 0.999),
                           B=1, seed=51),
    }


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def case_inputs(name, case):
    """Seeded inputs for a case (numpy fp32)."""
    s = case["seed"]
    B = case["B"]
    k = case["kind"]
    if k == "spade":
        return dict(x=fill.uniform((B, case["C"], case["H"], case["W"]), s * 100 + 1, -2, 2),
                    seg=fill.uniform((B, case["cond_nc"]) + tuple(case["cond_hw"]), s * 100 + 2))
    if k == "resblk":
        return dict(x=fill.uniform((B, case["fin"], case["H"], case["W"]), s * 100 + 1, -2, 2),
                    seg=fill.uniform((B, 3) + tuple(case["cond_hw"]), s * 100 + 2))
    if k == "painter":
        return dict(cond=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 1))
    if k == "paint":
        return dict(x=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 1),
                    m=fill.rect_mask(B, case["H"], case["W"], s * 100 + 2))
    if k == "disc_p":
        return dict(x=fill.uniform((B, 4, case["H"], case["W"]), s * 100 + 1))
    if k == "disc_fc":
        return dict(x=fill.uniform01((B, case["num_classes"], case["H"], case["W"]), s * 100 + 1).astype(np.float32))
    if k == "mstep":
        H, W = case["H"], case["W"]
        h, w = H // 4, W // 4
        d = {}
        for i, dom in enumerate(("r", "s")):
            d["x_" + dom] = fill.uniform((B, 3, H, W), s * 100 + 10 * i + 1)
            d["d_" + dom] = fill.uniform((B, 1, h, w), s * 100 + 10 * i + 2, 0.35, 6.95)
            d["s_" + dom] = (fill.uniform01((B, 1, h, w), s * 100 + 10 * i + 3) * 11).astype(np.int64).clip(0, 10)
            d["m_" + dom] = fill.rect_mask(B, H, W, s * 100 + 10 * i + 4)
        return d
    if k == "masker_losses":
        h, w = case["H"], case["W"]
        return dict(s_logits=fill.uniform((B, 11, h, w), s * 100 + 1, -2, 2),
                    s_target=(fill.uniform01((B, h, w), s * 100 + 2) * 11).astype(np.int64).clip(0, 10),
                    m_logits=fill.uniform((B, 1, 2 * h, 2 * w), s * 100 + 3, -3, 3),
                    m_target=fill.rect_mask(B, 2 * h, 2 * w, s * 100 + 4),
                    ground=fill.rect_mask(B, 2 * h, 2 * w, s * 100 + 5),
                    d_pred=fill.uniform((B, 1, h, w), s * 100 + 6, 0.2, 1.0),
                    d_out=fill.uniform((B, 1, 5, 7), s * 100 + 7, -1, 1),
                    depth_pred=fill.uniform((B, 1, h, w), s * 100 + 8, -1.0, 2.0),
                    depth_target=fill.uniform((B, 1, h, w), s * 100 + 9, 0.35, 6.95))
    if k == "gstep_p":
        m = fill.rect_mask(B, case["H"], case["W"], s * 100 + 3)
        if case.get("aux", {}).get("soft_mask"):
            m = (0.1 + 0.8 * m).astype(np.float32)
        return dict(x=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 1), m=m)
    if k == "fire":
        H, W = case["H"], case["W"]
        seg = fill.uniform((B, 11, H // 4, W // 4), s * 100 + 2, -1, 1)
        seg[:, case["sky_idx"], : H // 11, W // 20: W // 5] += 2.5      # a sky band in the upper part
        seg[:, case["sky_idx"], H // 16: H // 8, W // 8: W // 6] += 2.5  # a small detached blob
        seg[:, case["sky_idx"], 3 * H // 16:, : W // 10] += 2.5         # and one in the bottom third (cropped away)
        return dict(x=fill.uniform((B, 3, H, W), s * 100 + 1), seg=seg.astype(np.float16).astype(np.float32))
    if k == "hinge":
        # multiples of 1/8 in [-3, 3]: exact in fp16 / bf16 (the 16-bit path then sees the same logits), exact ties at +-1
        return {"p%d" % i: (np.round(fill.uniform((B, 1) + tuple(hw), s * 100 + i, -3, 3) * 8) / 8).astype(np.float32)
                for i, hw in enumerate(case["sizes"])}
    if k == "vgg":
        return dict(x=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 1),
                    fake=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 2),
                    m=fill.rect_mask(B, case["H"], case["W"], s * 100 + 3))
    if k == "dstep_p":
        return dict(x=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 1),
                    fake=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 2),
                    m=fill.rect_mask(B, case["H"], case["W"], s * 100 + 3))
    if k in ("masker", "infer", "cloudy", "maskspade"):
        return dict(x=fill.uniform((B, 3, case["H"], case["W"]), s * 100 + 1))
    if k == "extra_adam":
        d = {}
        for i, shp in enumerate(case["shapes"]):
            d["p%d" % i] = fill.uniform(shp, s * 100 + i)
            for st in range(case["steps"]):
                d["g%d_%d" % (i, st)] = fill.uniform(shp, s * 1000 + 10 * st + i, -0.5, 0.5)
        return d
    raise KeyError(k)


def summarize(y: np.ndarray) -> dict:
    """Compact statistics of a [B,C,H,W] output too large to commit in full."""
    yt = t(y).double()
    pooled = torch.nn.functional.avg_pool2d(yt, 8).float().numpy()
    h, w = y.shape[-2:]
    return dict(
        mean=yt.mean(dim=(2, 3)).float().numpy(),
        std=yt.std(dim=(2, 3), unbiased=False).float().numpy(),
        pooled8=pooled,
        crop_tl=y[..., :32, :32].copy(),
        crop_c=y[..., h // 2 - 16:h // 2 + 16, w // 2 - 16:w // 2 + 16].copy(),
        crop_br=y[..., -32:, -32:].copy(),
    )


# ------------------------------------------------------------------ reference builders
def _painter_opts(case):
    from oracle import ref_shim

    opts = ref_shim.default_opts()
    opts.gen.p.latent_dim = case["latent_dim"]
    opts.gen.p.spade_n_up = case["n_up"]
    return opts


def build_reference_module(case):
    """Instantiate the reference module for a case, loaded with the portable fill.  Returns (module, sd_np)."""
    from oracle import ref_shim

    k = case["kind"]
    if k == "spade":
        mod = ref_shim.ref("norms").SPADE("instance", 3, case["C"], case["cond_nc"])
    elif k == "resblk":
        mod = ref_shim.ref("blocks").SPADEResnetBlock(case["fin"], case["fout"], 3, True, "instance", 3)
    elif k in ("painter", "paint"):
        mod = ref_shim.ref("painter").PainterSpadeDecoder(_painter_opts(case))
    elif k == "disc_p":
        mod = ref_shim.ref("discriminator").define_D(
            input_nc=4, ndf=case["ndf"], n_layers=case["n_layers"], norm="instance", use_sigmoid=False,
            get_intermediate_features=True, num_D=case["num_D"])
    elif k == "disc_fc":
        mod = ref_shim.ref("discriminator").get_fc_discriminator(num_classes=case["num_classes"], use_norm=True)
    else:
        raise KeyError(k)
    shapes = {key: tuple(v.shape) for key, v in mod.state_dict().items()}
    sd_np = fill.fill_state_dict(shapes, case["seed"])
    mod.load_state_dict({key: t(v) for key, v in sd_np.items()})
    mod.eval()
    return mod, sd_np


def masker_generator(case):
    """The reference OmniGenerator (default config, masker tasks only) loaded with the portable fill, eval mode."""
    import contextlib
    import io

    from oracle import ref_shim

    opts = ref_shim.default_opts()
    opts.tasks = ["d", "s", "m"]
    with contextlib.redirect_stdout(io.StringIO()):
        G = ref_shim.ref("generator").create_generator(opts, "cpu", no_init=True)
    shapes = {key: tuple(v.shape) for key, v in G.state_dict().items()}
    sd_np = fill.fill_state_dict(shapes, case["seed"], gain=case["gain"])
    G.load_state_dict({key: t(v) for key, v in sd_np.items()})
    G.eval()
    return G, shapes


def run_reference_masker(name, case):
    """Trainer.infer_all's masker stage (trainer.py:272-287) on the reference generator."""
    G, shapes = masker_generator(case)
    x = t(case_inputs(name, case)["x"])
    hs, ws = case["H"] // 4, case["W"] // 4
    G.decoders["d"]._target_size = ws            # int, as find_target_size leaves it: depth.py:143 skips the re-sampling
    G.decoders["s"].set_target_size((hs, ws))
    with torch.no_grad():
        z = G.encode(x)
        d, z_depth = G.decoders["d"](z)
        s = G.decoders["s"](z, z_depth)
        m = G.mask(z=z, z_depth=z_depth)
    zh = z[0]
    out = {"d": d.numpy(), "s": s.numpy(), "m": m.numpy(),
           "z_high_mean": zh.mean(dim=(0, 2, 3)).numpy(), "z_high_std": zh.std(dim=(0, 2, 3)).numpy(),
           "z_high_crop": zh[:, :64, :8, :8].numpy().copy(), "z_depth_crop": z_depth[:, :64, :8, :8].numpy().copy()}
    for key, v in G.state_dict().items():
        if key.endswith("weight_u"):
            out["post." + key] = v.numpy().copy()
    return out


def reference_trainer(case):
    """The reference ``Trainer`` set up for inference (trainer.py:701-760) on CPU with the portable fill.  Patches
    (SURVEY 8c): ``Timer`` -> null context (it records CUDA events), ``compute_fire`` -> identity (kornia /
    torchvision arithmetic is not in the tree)."""
    import contextlib
    import io

    from oracle import ref_shim

    opts = ref_shim.default_opts()
    opts.tasks = ["d", "s", "m", "p"]
    opts.gen.p.latent_dim = case["latent_dim"]
    opts.gen.p.spade_n_up = case["n_up"]
    tr = ref_shim.ref("trainer")

    class NullTimer:
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False

    tr.Timer = NullTimer
    T = tr.Trainer(opts, device=torch.device("cpu"))
    with contextlib.redirect_stdout(io.StringIO()):
        T.setup(inference=True)
    shapes = {key: tuple(v.shape) for key, v in T.G.state_dict().items()}
    sd_np = fill.fill_state_dict(shapes, case["seed"], gain=case["gain"])
    T.G.load_state_dict({key: t(v) for key, v in sd_np.items()})
    T.G.eval()
    T.compute_fire = lambda x, seg_preds=None, **kw: x
    return T, shapes


def run_reference_infer(name, case):
    """``Trainer.infer_all`` (trainer.py:217-334) end to end; the float event tensors are captured on the way."""
    T, shapes = reference_trainer(case)
    x = t(case_inputs(name, case)["x"])
    cap = {}

    def capture(fn, key):
        def wrapped(*a, **kw):
            r = fn(*a, **kw)
            cap[key] = r.detach().clone()
            cap[key + "_kw"] = {k2: (v.detach().clone() if torch.is_tensor(v) else v) for k2, v in kw.items()}
            return r
        return wrapped

    T.compute_flood = capture(T.compute_flood, "flood")
    T.compute_smog = capture(T.compute_smog, "smog")
    res = T.infer_all(x, numpy=True, bin_value=case["bin_value"], return_masks=True)
    out = {"flood_u8": res["flood"], "smog_u8": res["smog"], "mask_u8": res["mask"],
           "flood": cap["flood"].numpy(), "smog": cap["smog"].numpy(),
           "m": cap["flood_kw"]["m"].numpy(), "s": cap["flood_kw"]["s"].numpy(), "d": cap["smog_kw"]["d"].numpy()}
    return out


def run_reference_cloudy(name, case):
    """``compute_flood(cloudy=True)`` -> ``OmniGenerator.paint_cloudy`` on the reference Trainer's generator, with the
    global torch RNG seeded right before the call (the Perlin angles are its only random draw)."""
    T, _ = reference_trainer(case)
    x = t(case_inputs(name, case)["x"])
    T.G.painter.set_latent_shape(x.shape, True)
    with torch.no_grad():
        z = T.G.encode(x)
        depth, z_depth = T.G.decoders["d"](z)
        seg = T.G.decoders["s"](z, z_depth)
        mask = T.G.mask(z=z, z_depth=z_depth)
        m_bin = (mask > case["bin_value"]).to(mask.dtype)
        torch.manual_seed(case["rng_seed"])
        flood = T.G.paint_cloudy(m_bin, x, seg, sky_idx=case["sky_idx"])      # what compute_flood(cloudy=True) calls
    sky = torch.argmax(torch.nn.functional.interpolate(seg, x.shape[-2:], mode="bilinear"), 1) == case["sky_idx"]
    return {"s": seg.numpy(), "m_bin": m_bin.numpy(), "flood": flood.numpy(),
            "sky_fraction": np.array([sky.float().mean().item()], dtype=np.float32)}


def maskspade_generator(case):
    """The reference OmniGenerator with ``gen.m.use_spade`` (MaskSpadeDecoder; its hard-coded ``.cuda()``,
    masker.py:196, is neutralised for the construction), masker tasks, eval mode, portable fill."""
    import contextlib
    import io

    from oracle import ref_shim

    opts = ref_shim.default_opts()
    opts.tasks = ["d", "s", "m"]
    opts.gen.m.use_spade = True
    orig = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **k: self
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            G = ref_shim.ref("generator").create_generator(opts, "cpu", no_init=True)
    finally:
        torch.nn.Module.cuda = orig
    shapes = {key: tuple(v.shape) for key, v in G.state_dict().items()}
    G.load_state_dict({key: t(v) for key, v in fill.fill_state_dict(shapes, case["seed"], gain=case["gain"]).items()})
    G.eval()
    return G, shapes


def run_reference_maskspade(name, case):
    """infer_all's masker stage (trainer.py:272-287) with the SPADE mask decoder: cond = make_m_cond(d, s, x)."""
    G, shapes = maskspade_generator(case)
    x = t(case_inputs(name, case)["x"])
    with torch.no_grad():
        z = G.encode(x)
        d, z_depth = G.decoders["d"](z)
        s = G.decoders["s"](z, z_depth)
        cond = G.make_m_cond(d, s, x)
        m = G.mask(z=z, cond=cond, z_depth=z_depth)
        logits = G.mask(z=z, cond=cond, z_depth=z_depth, sigmoid=False)      # second call: one more power iteration
    return {"d": d.numpy(), "s": s.numpy(), "cond": cond.numpy(), "m": m.numpy(), "logits2": logits.numpy()}


def run_reference_masker_losses(name, case):
    """The masker-side loss classes of the reference (losses.py) on seeded tensors: values and input gradients, composed
    as masker_s_loss / masker_m_loss compose them (trainer.py:1409-1616)."""
    from oracle import ref_shim

    L = ref_shim.ref("losses")
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    out = {}

    def record(key, loss, leaf):
        (g,) = torch.autograd.grad(loss, leaf)
        out[key] = loss.detach().numpy().reshape(1)
        out[key + ".grad"] = g.numpy()

    s = inp["s_logits"].clone().requires_grad_(True)
    record("crossent", L.CrossEntropy()(s, inp["s_target"]), s)
    s = inp["s_logits"].clone().requires_grad_(True)
    record("minent_v1", L.MinentLoss()(torch.softmax(s, dim=1)), s)
    s = inp["s_logits"].clone().requires_grad_(True)
    ent = L.prob_2_entropy(torch.softmax(s, dim=1)) * inp["d_pred"]
    out["entropy_dada"] = ent.detach().numpy()
    record("entropy_dada_sum", (ent * 0.37).sum(), s)
    m = inp["m_logits"].clone().requires_grad_(True)
    record("bce", torch.nn.BCEWithLogitsLoss()(m, inp["m_target"]), m)
    m = inp["m_logits"].clone().requires_grad_(True)
    record("tv", L.TVLoss()(torch.sigmoid(m)), m)
    m = inp["m_logits"].clone().requires_grad_(True)
    p = torch.sigmoid(m)
    record("minent_v2", L.MinentLoss(version=2, lambda_var=0.1)(torch.cat([p, 1 - p], dim=1)), m)
    out["gi"] = L.GroundIntersectionLoss()(torch.sigmoid(inp["m_logits"]), inp["ground"]).numpy().reshape(1)
    d = inp["d_out"].clone().requires_grad_(True)
    wgan = lambda x, y: -torch.mean(y * x + (1 - y) * (1 - x))      # losses.py:498-499  # noqa: E731
    record("advent_wgan_0", wgan(d, 0), d)
    d = inp["d_out"].clone().requires_grad_(True)
    record("advent_wgan_1", wgan(d, 1), d)
    dp = inp["depth_pred"].half().float().clone().requires_grad_(True)     # the values the 16-bit map can hold (ties!)
    record("sigm", L.SIGMLoss(0.5, device="cpu")(dp, inp["depth_target"]), dp)
    out["sigm.median_index"] = np.array([int(torch.median(dp.detach().flatten(), 0).indices)], dtype=np.int64)
    return out


def grad_subsample(key, g, n):
    """n entries of a gradient tensor at seeded pseudo-random positions (fixtures cannot hold 105 M-element gradients)."""
    flat = g.reshape(-1)
    if flat.numel() <= n:
        return flat.numpy().copy()
    idx = (fill.uniform01((n,), fill.key_seed(key, 4242)) * flat.numel()).astype(np.int64).clip(0, flat.numel() - 1)
    return flat[torch.from_numpy(idx)].numpy().copy()


def run_reference_mstep(name, case):
    """G side of the Masker step: ``get_masker_loss`` (trainer.py:1184-1254) restated on the reference's own modules
    (generator in train mode: batch-statistics BatchNorm) and loss classes, domains r then s, ADVENT discriminators
    frozen.  Gradients are stored as seeded sub-samples + norms."""
    import contextlib
    import io

    from oracle import ref_shim

    opts = ref_shim.default_opts()
    opts.tasks = ["d", "s", "m"]
    use_spade = bool(case.get("use_spade"))
    opts.gen.m.use_spade = use_spade
    L = ref_shim.ref("losses")
    disc = ref_shim.ref("discriminator")
    orig_cuda = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **k: self       # MaskSpadeDecoder's hard-coded .cuda() (masker.py:196)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            G = ref_shim.ref("generator").create_generator(opts, "cpu", no_init=True)
            D = disc.OmniDiscriminator(opts)
    finally:
        torch.nn.Module.cuda = orig_cuda
    for mod, seed in ((G, case["seed"]), (D, case["seed"] + 1)):
        shapes = {key: tuple(v.shape) for key, v in mod.state_dict().items()}
        mod.load_state_dict({key: t(v) for key, v in fill.fill_state_dict(shapes, seed, gain=case["gain"]).items()})
    G.train()
    D.train()
    for p in D.parameters():
        p.requires_grad = False
    hs, ws = case["H"] // 4, case["W"] // 4
    G.decoders["d"]._target_size = ws
    G.decoders["s"].set_target_size((hs, ws))
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    lam = opts.train.lambdas
    crossent, minent1, minent2 = L.CrossEntropy(), L.MinentLoss(), L.MinentLoss(version=2, lambda_var=lam.advent.ent_var)
    tv, bce, gi = L.TVLoss(), torch.nn.BCEWithLogitsLoss(), L.GroundIntersectionLoss()
    adv_s = L.ADVENTAdversarialLoss(opts, gan_type=opts.dis.s.gan_type)
    adv_m = L.ADVENTAdversarialLoss(opts, gan_type=opts.dis.m.gan_type)
    sigm = L.SIGMLoss(lam.G.d.gml, device="cpu")
    terms = {}
    total = 0
    for dom in ("r", "s"):
        x = inp["x_" + dom]
        z = G.encode(x)
        d_pred, z_depth = G.decoders["d"](z)
        l = sigm(d_pred, inp["d_" + dom]) * lam.G.d.main
        terms["d." + dom] = l.detach()
        if dom == "s":                                   # real-domain depth loss is computed and discarded (trainer.py:1403-1405)
            total = total + l
        s_pred = G.decoders["s"](z, z_depth)
        if dom == "s":
            l = crossent(s_pred, inp["s_" + dom].squeeze(1)) * lam.G.s.crossent
            terms["s.crossent.s"] = l.detach(); total = total + l
        else:
            sm = torch.softmax(s_pred, dim=1)
            l = minent1(sm) * lam.G.s.minent
            terms["s.minent.r"] = l.detach(); total = total + l
            l = adv_s(sm, 0, D["s"]["Advent"], d_pred.detach()) * lam.G.s.advent
            terms["s.advent.r"] = l.detach(); total = total + l
        cond = G.make_m_cond(d_pred, s_pred, x) if use_spade else None       # trainer.py:1233-1238 (not detached)
        logits = G.decoders["m"](z, cond=cond, z_depth=z_depth)
        p = torch.sigmoid(logits)
        prob = torch.cat([p, 1 - p], dim=1)
        l = tv(p) * lam.G.m.tv
        terms["m.tv." + dom] = l.detach(); total = total + l
        if dom == "s":
            l = bce(logits, inp["m_" + dom]) * lam.G.m.bce
            terms["m.bce.s"] = l.detach(); total = total + l
        else:
            l = gi(p, inp["m_" + dom]) * lam.G.m.gi
            terms["m.gi.r"] = l.detach(); total = total + l
            l = minent2(prob) * lam.advent.ent_main
            terms["m.minent.r"] = l.detach(); total = total + l
            l = adv_m(prob, 0, D["m"]["Advent"], None) * lam.advent.adv_main
            terms["m.advent.r"] = l.detach(); total = total + l
    total.backward()
    out = {"loss": total.detach().numpy().reshape(1)}
    for k2, v in terms.items():
        out["term." + k2] = v.numpy().reshape(1)
    for key, p in G.named_parameters():
        if p.requires_grad and p.grad is not None:
            out["gsub." + key] = grad_subsample(key, p.grad, case["sub"])
            out["gnorm." + key] = np.array([p.grad.norm().item()], dtype=np.float32)
    sd = G.state_dict()
    for key in ("encoder.bn1.running_mean", "encoder.layer3.5.bn2.running_var", "decoders.s.aspp.conv2.bn.running_mean",
                "decoders.d.enc4_2.norm.running_var"):
        out["post." + key] = sd[key].numpy().copy()
    for key in sd:
        if key.endswith("weight_u") and key.startswith("decoders.m."):
            out["post." + key] = sd[key].numpy().copy()
        if use_spade and key.startswith("decoders.m.") and (key.endswith("running_mean") or key.endswith("running_var")):
            out["post." + key] = sd[key].numpy().copy()
    return out


def run_reference_dstep(name, case):
    """The Painter branch of ``Trainer.get_D_loss`` (trainer.py:1073-1107) with the reference's own modules and
    losses (``GANLoss`` as built by ``get_losses``: BCE form, here soft_shift = flip_prob = 0), then
    ``d_loss.backward()``: loss value + gradient of every trainable D parameter."""
    from oracle import ref_shim

    disc = ref_shim.ref("discriminator")
    losses = ref_shim.ref("losses")
    tutils = ref_shim.ref("tutils")
    D = disc.define_D(input_nc=4, ndf=case["ndf"], n_layers=case["n_layers"], norm="instance", use_sigmoid=False,
                      get_intermediate_features=True, num_D=case["num_D"])
    shapes = {key: tuple(v.shape) for key, v in D.state_dict().items()}
    sd_np = fill.fill_state_dict(shapes, case["seed"])
    D.load_state_dict({key: t(v) for key, v in sd_np.items()})
    D.train()
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    gan = losses.GANLoss(use_lsgan=False, soft_shift=0.0, flip_prob=0.0)
    real_cat = torch.cat([inp["m"], inp["x"]], axis=1)
    fake_cat = torch.cat([inp["m"], inp["fake"]], axis=1)
    real_fake_d = D(torch.cat([real_cat, fake_cat], dim=0))
    real_d, fake_d = tutils.divide_pred(real_fake_d)
    loss = gan(fake_d, False, True)
    loss += gan(real_d, True, True)
    loss.backward()
    out = {"loss": loss.detach().numpy().reshape(1)}
    for key, p in D.named_parameters():
        if p.requires_grad:
            out["grad." + key] = p.grad.numpy().copy()
        if key.endswith("weight_u"):
            out["post." + key] = p.data.numpy().copy()
    return out


def run_reference_gstep(name, case):
    """The G side of the Painter step with the reference's own modules: ``OmniGenerator.paint`` around the reference
    Painter, the reference D, ``GANLoss`` / ``FeatMatchLoss`` as built by ``get_losses``, the single-discriminator
    branch of ``get_painter_loss`` (trainer.py:1359-1385) with the VGG term off, D frozen, ``backward()``."""
    from oracle import ref_shim

    gen = ref_shim.ref("generator")
    disc = ref_shim.ref("discriminator")
    losses = ref_shim.ref("losses")
    tutils = ref_shim.ref("tutils")
    painter, _ = build_reference_module(dict(case, kind="painter"))
    painter.train()
    G = gen.OmniGenerator.__new__(gen.OmniGenerator)
    torch.nn.Module.__init__(G)
    G.opts = _painter_opts(case)
    G.painter = painter
    D = disc.define_D(input_nc=4, ndf=case["ndf"], n_layers=case["n_layers"], norm="instance", use_sigmoid=False,
                      get_intermediate_features=True, num_D=case["num_D"])
    dshapes = {key: tuple(v.shape) for key, v in D.state_dict().items()}
    D.load_state_dict({key: t(v) for key, v in fill.fill_state_dict(dshapes, case["seed"] + 1).items()})
    D.train()
    for p in D.parameters():
        p.requires_grad = False
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    x, m = inp["x"], inp["m"]
    painter.set_latent_shape(tuple(x.shape), True)
    aux = case.get("aux", {})
    gan, fm = losses.GANLoss(use_lsgan=bool(aux.get("lsgan"))), losses.FeatMatchLoss()
    fake_flooded = G.paint(m, x)
    real_cat = torch.cat([m, x], axis=1)
    fake_cat = torch.cat([m, fake_flooded], axis=1)
    real_fake_d = D(torch.cat([real_cat, fake_cat], dim=0))
    real_d, fake_d = tutils.divide_pred(real_fake_d)
    l_gan = gan(fake_d, True, False)
    l_fm = fm(real_d, fake_d) * 10
    loss = l_gan + l_fm
    extra = {}
    if aux:                                                    # the reference's own classes, get_painter_loss's expressions
        extra["tv"] = losses.TVLoss()(fake_flooded * m) * aux["tv"]                              # trainer.py:1289-1293
        extra["context"] = losses.ContextLoss()(fake_flooded, x, m) * aux["context"]             # :1298-1302
        extra["reconstruction"] = losses.ReconstructionLoss()(fake_flooded, x, m) * aux["reconstruction"]   # :1307-1311
        loss = loss + sum(extra.values())
    loss.backward()
    out = {"loss": loss.detach().numpy().reshape(1), "gan": l_gan.detach().numpy().reshape(1),
           "featmatch": l_fm.detach().numpy().reshape(1), "fake": fake_flooded.detach().numpy()}
    out.update({k2: v.detach().numpy().reshape(1) for k2, v in extra.items()})
    for key, p in painter.named_parameters():
        if p.requires_grad:
            out["grad." + key] = p.grad.numpy().copy()
    return out


def run_reference_gstep_local(name, case):
    """``dis.p.use_local_discriminator``: the reference's ``OmniDiscriminator`` builds D["p"] = {"global", "local"}, two
    3-channel multiscale discriminators (discriminator.py:242-324).  G side = trainer.py:1323-1358 (GAN terms of both,
    scaled by lambdas.G.p.gan; feature matching on the global one only), D side = trainer.py:1085-1099 on the detached fake
    (real / fake in separate calls, the local one on ``* m``), each with its own ``backward()``."""
    from oracle import ref_shim

    gen = ref_shim.ref("generator")
    disc = ref_shim.ref("discriminator")
    losses = ref_shim.ref("losses")
    painter, _ = build_reference_module(dict(case, kind="painter"))
    painter.train()
    G = gen.OmniGenerator.__new__(gen.OmniGenerator)
    torch.nn.Module.__init__(G)
    G.opts = _painter_opts(case)
    G.painter = painter
    opts = ref_shim.default_opts()
    opts.tasks = ["p"]
    opts.dis.p.use_local_discriminator = True
    opts.dis.p.ndf, opts.dis.p.n_layers, opts.dis.p.num_D = case["ndf"], case["n_layers"], case["num_D"]
    D = disc.OmniDiscriminator(opts)
    assert set(D["p"].keys()) == {"global", "local"}
    for i, which in enumerate(("global", "local")):
        shapes = {key: tuple(v.shape) for key, v in D["p"][which].state_dict().items()}
        D["p"][which].load_state_dict({key: t(v) for key, v in fill.fill_state_dict(shapes, case["seed"] + 1 + i).items()})
    D.train()
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    x, m = inp["x"], inp["m"]
    painter.set_latent_shape(tuple(x.shape), True)
    gan, fm = losses.GANLoss(use_lsgan=False, soft_shift=0.0, flip_prob=0.0), losses.FeatMatchLoss()   # get_losses, :384-388
    lam = case["local"]["lambda_gan"]
    out = {}
    # ---- G side (D frozen, as update_G leaves it)
    for p in D.parameters():
        p.requires_grad = False
    fake_flooded = G.paint(m, x)
    fake_d_global = D["p"]["global"](fake_flooded)
    fake_d_local = D["p"]["local"](fake_flooded * m)
    real_d_global = D["p"]["global"](x)
    l_gan = gan(fake_d_global, True, False)
    l_gan = l_gan + gan(fake_d_local, True, False)
    l_gan = l_gan * lam
    l_fm = fm(real_d_global, fake_d_global) * 10
    loss = l_gan + l_fm
    aux, extra = case.get("aux", {}), {}
    if aux:                                                    # trainer.py:1289-1315 come before the pair's terms
        extra["tv"] = losses.TVLoss()(fake_flooded * m) * aux["tv"]
        extra["context"] = losses.ContextLoss()(fake_flooded, x, m) * aux["context"]
        extra["reconstruction"] = losses.ReconstructionLoss()(fake_flooded, x, m) * aux["reconstruction"]
        loss = loss + sum(extra.values())
    loss.backward()
    out.update({"loss": loss.detach().numpy().reshape(1), "gan": l_gan.detach().numpy().reshape(1),
                "featmatch": l_fm.detach().numpy().reshape(1), "fake": fake_flooded.detach().numpy()})
    out.update({k2: v.detach().numpy().reshape(1) for k2, v in extra.items()})
    for key, p in painter.named_parameters():
        if p.requires_grad:
            out["grad." + key] = p.grad.numpy().copy()
    for key, v in D.state_dict().items():
        if key.endswith("weight_u"):
            out["post_g." + key] = v.numpy().copy()
    # ---- D side, on the same (detached) fake, continuing from the state the G side left (u / v advanced)
    for p in D.parameters():
        p.requires_grad = True
    fake = fake_flooded.detach()
    gan_d = losses.GANLoss(use_lsgan=False, soft_shift=0.0, flip_prob=0.0)
    g_loss = gan_d(D["p"]["global"](fake), False, True) + gan_d(D["p"]["global"](x), True, True)
    l_loss = gan_d(D["p"]["local"](fake * m), False, True) + gan_d(D["p"]["local"](x * m), True, True)
    (g_loss + l_loss).backward()
    out["d.global"] = g_loss.detach().numpy().reshape(1)
    out["d.local"] = l_loss.detach().numpy().reshape(1)
    for which in ("global", "local"):
        for key, p in D["p"][which].named_parameters():
            if p.grad is not None and not key.endswith(("weight_u", "weight_v")):
                out["dgrad.%s.%s" % (which, key)] = p.grad.numpy().copy()
    return out


def reference_vgg_loss(case):
    """The reference's ``VGGLoss`` (losses.py:337-350) around its own ``Vgg19`` (losses.py:304-334).  ``Vgg19`` asks
    torchvision for ``models.vgg19(pretrained=True).features``; torchvision is not installed, so the shim's dummy
    ``models`` gets a ``vgg19`` that returns torchvision's published configuration E as a plain ``nn.Sequential``
    (conv3x3-ReLU(inplace) stacks with 2x2 max pools) -- the slicing, the five taps, the weights (1/32 ... 1) and the
    L1 criterion are the reference's code.  Parameters: the portable fill."""
    from oracle import cpu_ref, ref_shim

    losses = ref_shim.ref("losses")

    def vgg19(pretrained=True, **kw):
        mods, cin = [], 3
        for v in cpu_ref.VGG19_E + ("M",):        # torchvision's features has 37 entries; the reference reads [0, 30)
            if v == "M":
                mods.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                mods += [torch.nn.Conv2d(cin, v, kernel_size=3, padding=1), torch.nn.ReLU(inplace=True)]
                cin = v
        holder = torch.nn.Module()
        holder.features = torch.nn.Sequential(*mods)
        return holder

    losses.models.vgg19 = vgg19
    crit = losses.VGGLoss("cpu")
    shapes = {key: tuple(v.shape) for key, v in crit.vgg.state_dict().items()}
    assert shapes == cpu_ref.vgg19_shapes(), "reference Vgg19 layout changed"
    crit.vgg.load_state_dict({key: t(v) for key, v in fill.fill_state_dict(shapes, case["seed"], gain=case["gain"]).items()})
    return crit, shapes


def run_reference_vgg(name, case):
    """VGG term of ``get_painter_loss`` (trainer.py:1276-1287) with the reference's ``vgg_preprocess`` (its hard-coded
    ``.cuda()``, tutils.py:422, patched to the identity), on the pasted image of ``OmniGenerator.paint``
    (generator.py:295-296); value, per-tap L1 terms and the gradient w.r.t. the painter's (pre-paste) output."""
    from oracle import ref_shim

    tutils = ref_shim.ref("tutils")
    crit, _ = reference_vgg_loss(case)
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    x, m = inp["x"], inp["m"]
    fake = inp["fake"].clone().requires_grad_(True)
    saved = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        fake_flooded = x * (1.0 - m) + fake * m
        a, b = tutils.vgg_preprocess(fake_flooded * m), tutils.vgg_preprocess(x * m)
        loss = crit(a, b) * case["lambda_vgg"]
        loss.backward()
        with torch.no_grad():
            fa, fb = crit.vgg(a), crit.vgg(b)
    finally:
        torch.Tensor.cuda = saved
    return {"loss": loss.detach().numpy().reshape(1), "dfake": fake.grad.numpy().copy(),
            "terms": np.array([(u - v).abs().mean().item() for u, v in zip(fa, fb)], dtype=np.float32),
            "feat_absmean": np.array([u.abs().mean().item() for u in fa], dtype=np.float32),
            "pre_fake": a.detach().numpy()}


def run_reference_fire(name, case):
    """The reference's own ``fire.add_fire`` (fire.py:70-133) -- normalisation to [0, 255], the warm shift, the sky mask
    from the segmentation (``retrieve_sky_mask``, bottom third cropped), nearest resize, ``increase_sky_mask`` (18 %),
    the blurred-mask paste with transparency 200 and the dummy range pixels are the reference's code.  Its three
    third-party calls are NOT in the tree (kornia 0.5.10, torchvision 0.8) and are bound to the formulas their
    documentation gives: ``adjust_brightness(img, f) = clamp(f * img)`` and ``adjust_contrast(img, f) =
    clamp(f * img + (1 - f) * mean(gray(img)))`` on uint8 with gray = 0.2989 R + 0.587 G + 0.114 B truncated to uint8;
    ``get_gaussian_kernel2d`` = outer product of sum-normalised 1-D Gaussians centred at k // 2; ``filter2d`` =
    reflect-pad k // 2, depthwise correlation -- here as a DIRECT 2-D correlation (no separable shortcut).  Only those
    three formulas stay unpinned.  ``random.randint(100, 150)`` (fire.py:115) is seeded and recorded."""
    import random

    from oracle import cpu_ref, ref_shim

    fire = ref_shim.ref("fire")

    def gaussian_kernel2d(kernel_size, sigma, *a, **k):
        gy = cpu_ref._kornia_gaussian_1d(kernel_size[0], sigma[0])
        gx = cpu_ref._kornia_gaussian_1d(kernel_size[1], sigma[1])
        return torch.outer(gy, gx)

    def filter2d(inp, kernel, border_type="reflect", *a, **k):
        assert border_type == "reflect" and kernel.dim() == 3 and kernel.shape[0] == 1
        kh, kw = kernel.shape[-2:]
        pad = F_.pad(inp.double(), (kw // 2, kw // 2, kh // 2, kh // 2), mode="reflect")
        c = inp.shape[1]
        w = kernel.double().reshape(1, 1, kh, kw).expand(c, 1, kh, kw)
        return F_.conv2d(pad, w, groups=c).float()

    import torch.nn.functional as F_
    fire.adjust_contrast = lambda img, contrast_factor: cpu_ref._tv_adjust_contrast_u8(img, contrast_factor)
    fire.adjust_brightness = lambda img, brightness_factor: cpu_ref._tv_adjust_brightness_u8(img, brightness_factor)
    fire.filter2d = filter2d
    fire.kornia.filters.kernels.get_gaussian_kernel2d = gaussian_kernel2d
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    opts = ref_shim.default_opts()
    random.seed(case["rng_seed"])
    green = random.randint(100, 150)
    random.seed(case["rng_seed"])
    with torch.no_grad():
        y = fire.add_fire(inp["x"].clone(), inp["seg"], opts.events.fire)
    assert (y == y.round()).all() and y.min() >= 0 and y.max() <= 255
    return {"y_u8": y.numpy().astype(np.uint8), "green": np.array([green], dtype=np.int64)}


def run_reference_hinge(name, case):
    """The reference's ``HingeLoss`` (losses.py:550-593) on a 3-scale list of lists: D-real, D-fake and G values and
    the gradients w.r.t. every scale's prediction."""
    from oracle import ref_shim

    crit = ref_shim.ref("losses").HingeLoss()
    inp = case_inputs(name, case)
    out = {}
    for tag, real, for_d in (("d_real", True, True), ("d_fake", False, True), ("g", True, False)):
        preds = [t(inp["p%d" % i]).clone().requires_grad_(True) for i in range(len(case["sizes"]))]
        loss = crit([[p * 0, p] for p in preds], real, for_d)
        loss.backward()
        out[tag] = loss.detach().numpy().reshape(1)
        for i, p in enumerate(preds):
            out["%s.grad%d" % (tag, i)] = p.grad.numpy().copy()
    return out


def run_reference_extra_adam(name, case):
    """4-call trajectory extrapolation/step/extrapolation/step of the reference's ExtraAdam (optim.py:200-291)."""
    from oracle import ref_shim

    optim = ref_shim.ref("optim")
    inp = case_inputs(name, case)
    params = [torch.nn.Parameter(t(inp["p%d" % i]).clone()) for i in range(len(case["shapes"]))]
    opt = optim.ExtraAdam(params, lr=case["lr"], betas=tuple(case["betas"]))
    out = {}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for st in range(case["steps"]):
            for i, p in enumerate(params):
                p.grad = t(inp["g%d_%d" % (i, st)]).clone()
            if st % 2 == 0:
                opt.extrapolation()
            else:
                opt.step()
            for i, p in enumerate(params):
                out["p%d_after%d" % (i, st)] = p.data.numpy().copy()
    for i, p in enumerate(params):
        out["m%d" % i] = opt.state[p]["exp_avg"].numpy().copy()
        out["v%d" % i] = opt.state[p]["exp_avg_sq"].numpy().copy()
    return out


def run_reference(name, case):
    """Run the real reference on the seeded inputs; returns dict of numpy outputs."""
    from oracle import ref_shim

    if case["kind"] == "extra_adam":
        return run_reference_extra_adam(name, case)
    if case["kind"] == "masker":
        return run_reference_masker(name, case)
    if case["kind"] == "infer":
        return run_reference_infer(name, case)
    if case["kind"] == "cloudy":
        return run_reference_cloudy(name, case)
    if case["kind"] == "maskspade":
        return run_reference_maskspade(name, case)
    if case["kind"] == "masker_losses":
        return run_reference_masker_losses(name, case)
    if case["kind"] == "mstep":
        return run_reference_mstep(name, case)
    if case["kind"] == "dstep_p":
        return run_reference_dstep(name, case)
    if case["kind"] == "gstep_p" and case.get("local"):
        return run_reference_gstep_local(name, case)
    if case["kind"] == "gstep_p":
        return run_reference_gstep(name, case)
    if case["kind"] == "vgg":
        return run_reference_vgg(name, case)
    if case["kind"] == "fire":
        return run_reference_fire(name, case)
    if case["kind"] == "hinge":
        return run_reference_hinge(name, case)
    mod, _ = build_reference_module(case)
    inp = {k2: t(v) for k2, v in case_inputs(name, case).items()}
    out = {}
    k = case["kind"]
    with torch.no_grad():
        if k in ("spade", "resblk"):
            out["y"] = mod(inp["x"], inp["seg"]).numpy()
        elif k == "painter":
            mod.set_latent_shape(tuple(inp["cond"].shape), True)
            y = mod(None, inp["cond"]).numpy()
            if case["full"]:
                out["y"] = y
            else:
                out.update({"y_" + a: b for a, b in summarize(y).items()})
        elif k == "paint":
            # OmniGenerator.paint (generator.py:279-297) around the reference painter, no_z / paste defaults.
            gen = ref_shim.ref("generator")
            G = gen.OmniGenerator.__new__(gen.OmniGenerator)
            torch.nn.Module.__init__(G)
            G.opts = _painter_opts(case)
            G.painter = mod
            mod.set_latent_shape(tuple(inp["x"].shape), True)
            out["y"] = G.paint(inp["m"], inp["x"]).numpy()
        elif k == "disc_p":
            res = mod(inp["x"])
            for i, scale in enumerate(res):
                for j, f in enumerate(scale):
                    out["d%d_%d" % (i, j)] = f.numpy()
        elif k == "disc_fc":
            out["y"] = mod(inp["x"]).numpy()
    # post-forward spectral-norm state (the reference mutates u/v on every forward, norms.py:100-112)
    for key, v in mod.state_dict().items():
        if key.endswith("weight_u"):
            out["post." + key] = v.numpy().copy()
    return out


def main():
    from oracle import ref_shim

    if not ref_shim.available():
        sys.exit("make_golden needs /root/reference (dev container only)")
    GOLDEN_DIR.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])           # optional: regenerate just the named cases
    manifest = json.loads((GOLDEN_DIR / "manifest.json").read_text()) if only else {}
    for name, case in golden_cases().items():
        if only and name not in only:
            continue
        out = run_reference(name, case)
        path = GOLDEN_DIR / (name + ".npz")
        np.savez_compressed(path, **out)
        manifest[name] = dict(case=case, keys=sorted(out), bytes=path.stat().st_size)
        print("%-14s %8d B  %s" % (name, path.stat().st_size, ", ".join(sorted(out)[:6])))
    (GOLDEN_DIR / "manifest.json").write_text(json.dumps(manifest, indent=1, default=list))
    # state-dict layout of the reference's default generator (masker part) -- data for the layout tests
    _, shapes = masker_generator(golden_cases()["masker_small"])
    (GOLDEN_DIR / "generator_masker_shapes.json").write_text(json.dumps({k: list(v) for k, v in shapes.items()}))
    _, shapes = maskspade_generator(golden_cases()["maskspade_small"])
    shapes = {k: v for k, v in shapes.items() if k.startswith("decoders.m.")}
    (GOLDEN_DIR / "generator_maskspade_shapes.json").write_text(json.dumps({k: list(v) for k, v in shapes.items()}))


if __name__ == "__main__":
    main()
