"""Shared helpers for the tests: run the oracle (oracle.cpu_ref) on a golden case."""
from pathlib import Path

import numpy as np
import torch

from climategan_amd import fill
from oracle import cpu_ref
from oracle.make_golden import case_inputs, golden_cases, summarize

GOLDEN = Path(__file__).resolve().parent / "golden"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def module_shapes(case):
    """State-dict shapes of the reference module for a case, restated (no reference import)."""
    k = case["kind"]
    if k == "spade":
        return spade_shapes("", case["C"], case["cond_nc"])
    if k == "resblk":
        return resblk_shapes("", case["fin"], case["fout"], 3)
    if k in ("painter", "paint"):
        return painter_shapes(case["latent_dim"], case["n_up"])
    if k == "disc_p":
        return disc_p_shapes(4, case["ndf"], case["n_layers"], case["num_D"])
    if k == "disc_fc":
        return disc_fc_shapes(case["num_classes"])
    if k == "gstep_p":
        return painter_shapes(case["latent_dim"], case["n_up"])
    if k == "dstep_p":
        return disc_p_shapes(4, case["ndf"], case["n_layers"], case["num_D"])
    if k in ("extra_adam", "masker", "infer", "cloudy", "maskspade", "masker_losses", "mstep"):
        return {}
    raise KeyError(k)


def _p(prefix, k):
    return (prefix + "." + k) if prefix else k


def spade_shapes(prefix, C, cond_nc, nhidden=128, ks=3):
    return {
        _p(prefix, "mlp_shared.0.weight"): (nhidden, cond_nc, ks, ks),
        _p(prefix, "mlp_shared.0.bias"): (nhidden,),
        _p(prefix, "mlp_gamma.weight"): (C, nhidden, ks, ks),
        _p(prefix, "mlp_gamma.bias"): (C,),
        _p(prefix, "mlp_beta.weight"): (C, nhidden, ks, ks),
        _p(prefix, "mlp_beta.bias"): (C,),
    }


def sn_conv_shapes(prefix, cin, cout, k, bias=True):
    d = {}
    if bias:
        d[_p(prefix, "module.bias")] = (cout,)
    d[_p(prefix, "module.weight_u")] = (cout,)
    d[_p(prefix, "module.weight_v")] = (cin * k * k,)
    d[_p(prefix, "module.weight_bar")] = (cout, cin, k, k)
    return d


def resblk_shapes(prefix, fin, fout, cond_nc):
    fmid = min(fin, fout)
    d = {}
    d.update(sn_conv_shapes(_p(prefix, "conv_0"), fin, fmid, 3))
    d.update(sn_conv_shapes(_p(prefix, "conv_1"), fmid, fout, 3))
    if fin != fout:
        d.update(sn_conv_shapes(_p(prefix, "conv_s"), fin, fout, 1, bias=False))
    d.update(spade_shapes(_p(prefix, "norm_0"), fin, cond_nc))
    d.update(spade_shapes(_p(prefix, "norm_1"), fmid, cond_nc))
    if fin != fout:
        d.update(spade_shapes(_p(prefix, "norm_s"), fin, cond_nc))
    return d


def painter_shapes(latent_dim, n_up):
    d = {"fc.weight": (latent_dim, 3, 3, 3), "fc.bias": (latent_dim,)}
    for b in ("head_0", "G_middle_0", "G_middle_1"):
        d.update(resblk_shapes(b, latent_dim, latent_dim, 3))
    for i in range(n_up - 2):
        d.update(resblk_shapes("up_spades.%d" % i, latent_dim // 2 ** i, latent_dim // 2 ** (i + 1), 3))
    fnc = latent_dim // 2 ** (n_up - 2)
    d.update(resblk_shapes("final_spade", fnc, fnc, 3))
    d["conv_img.weight"] = (3, fnc, 3, 3)
    d["conv_img.bias"] = (3,)
    return d


def disc_p_shapes(input_nc, ndf, n_layers, num_D):
    d = {}
    for i in range(num_D):
        pre = "discriminator_%d" % i
        d.update(sn_conv_shapes(pre + ".model0.0", input_nc, ndf, 4))
        nf = 1
        for n in range(1, n_layers):
            nfp, nf = nf, min(2 ** n, 8)
            d.update(sn_conv_shapes(pre + ".model%d.0" % n, ndf * nfp, ndf * nf, 4))
        nfp, nf = nf, min(2 ** n_layers, 8)
        d.update(sn_conv_shapes(pre + ".model%d.0" % n_layers, ndf * nfp, ndf * nf, 4))
        d.update(sn_conv_shapes(pre + ".model%d.0" % (n_layers + 1), ndf * nf, 1, 4))
    return d


def disc_fc_shapes(num_classes, ndf=64):
    chans = [num_classes, ndf, ndf * 2, ndf * 4, ndf * 8, 1]
    d = {}
    for i, idx in enumerate((0, 2, 4, 6, 8)):
        d.update(sn_conv_shapes(str(idx), chans[i], chans[i + 1], 4))
    return d


def case_state_dict(case, dtype=torch.float32):
    sd = fill.fill_state_dict(module_shapes(case), case["seed"])
    return {k: t(v).to(dtype) if v.dtype != np.int64 else t(v) for k, v in sd.items()}


def run_oracle_extra_adam(name, case):
    inp = {k: t(v) for k, v in case_inputs(name, case).items()}
    n = len(case["shapes"])
    opt = cpu_ref.ExtraAdamRef([inp["p%d" % i].clone() for i in range(n)], lr=case["lr"], betas=tuple(case["betas"]))
    out = {}
    for st in range(case["steps"]):
        grads = [inp["g%d_%d" % (i, st)] for i in range(n)]
        (opt.extrapolation if st % 2 == 0 else opt.step)(grads)
        for i in range(n):
            out["p%d_after%d" % (i, st)] = opt.params[i].numpy().copy()
    for i in range(n):
        out["m%d" % i] = opt.state[i]["exp_avg"].numpy().copy()
        out["v%d" % i] = opt.state[i]["exp_avg_sq"].numpy().copy()
    return out


def masker_shapes():
    import json

    return {k: tuple(v) for k, v in json.loads((GOLDEN / "generator_masker_shapes.json").read_text()).items()}


def masker_state_dict(case):
    sd = fill.fill_state_dict(masker_shapes(), case["seed"], gain=case["gain"])
    return {k: t(v) for k, v in sd.items()}


def run_oracle_masker(name, case):
    sd = masker_state_dict(case)
    x = t(case_inputs(name, case)["x"])
    with torch.no_grad():
        r = cpu_ref.masker_forward(sd, x, (case["H"] // 4, case["W"] // 4))
    zh = r["z_high"]
    out = {"d": r["d"].numpy(), "s": r["s"].numpy(), "m": r["m"].numpy(),
           "z_high_mean": zh.mean(dim=(0, 2, 3)).numpy(), "z_high_std": zh.std(dim=(0, 2, 3)).numpy(),
           "z_high_crop": zh[:, :64, :8, :8].numpy().copy(), "z_depth_crop": r["z_depth"][:, :64, :8, :8].numpy().copy()}
    for key, v in sd.items():
        if key.endswith("weight_u"):
            out["post." + key] = v.numpy().copy()
    return out


def infer_shapes(case):
    """Full generator (masker + painter) state-dict shapes of an "infer" case."""
    shapes = dict(masker_shapes())
    shapes.update({"painter." + k: v for k, v in painter_shapes(case["latent_dim"], case["n_up"]).items()})
    return shapes


def infer_state_dict(case):
    sd = fill.fill_state_dict(infer_shapes(case), case["seed"], gain=case["gain"])
    return {k: t(v) for k, v in sd.items()}


def run_oracle_infer(name, case):
    sd = infer_state_dict(case)
    x = t(case_inputs(name, case)["x"])
    with torch.no_grad():
        r = cpu_ref.infer_all_flood(sd, x, case["n_up"], case["bin_value"])
    return {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in r.items()}


def run_oracle_dstep(name, case):
    sd = case_state_dict(case)
    inp = {k: t(v) for k, v in case_inputs(name, case).items()}
    loss, grads = cpu_ref.painter_d_step(sd, inp["m"], inp["x"], inp["fake"], case["num_D"], case["n_layers"])
    out = {"loss": loss.numpy().reshape(1)}
    for k, g in grads.items():
        out["grad." + k] = g.numpy()
    for k, v in sd.items():
        if k.endswith("weight_u"):
            out["post." + k] = v.numpy().copy()
    return out


def gstep_d_state_dict(case):
    shapes = disc_p_shapes(4, case["ndf"], case["n_layers"], case["num_D"])
    return {k: t(v) for k, v in fill.fill_state_dict(shapes, case["seed"] + 1).items()}


def run_oracle_gstep(name, case):
    sd_p = case_state_dict(case)
    sd_d = gstep_d_state_dict(case)
    inp = {k: t(v) for k, v in case_inputs(name, case).items()}
    z_h, z_w = case["H"] // 2 ** case["n_up"], case["W"] // 2 ** case["n_up"]
    with torch.no_grad():
        fake = cpu_ref.paint({k: v.clone() for k, v in sd_p.items()}, inp["m"], inp["x"], z_h, z_w)
    loss, grads, terms = cpu_ref.painter_g_step(sd_p, sd_d, inp["m"], inp["x"], z_h, z_w, case["num_D"],
                                                case["n_layers"])
    out = {"loss": loss.numpy().reshape(1), "gan": terms["gan"].numpy().reshape(1),
           "featmatch": terms["featmatch"].numpy().reshape(1), "fake": fake.numpy()}
    for k, g in grads.items():
        out["grad." + k] = g.numpy()
    return out


def run_oracle_cloudy(name, case):
    import math

    gold = load_golden(name)
    sd = infer_state_dict(case)
    x = t(case_inputs(name, case)["x"])
    torch.manual_seed(case["rng_seed"])
    angles = 2 * math.pi * torch.rand(9, 9)
    z_h, z_w = case["H"] // 2 ** case["n_up"], case["W"] // 2 ** case["n_up"]
    with torch.no_grad():
        flood = cpu_ref.paint_cloudy(cpu_ref.sub(sd, "painter"), t(gold["m_bin"]), x, t(gold["s"]), z_h, z_w, angles,
                                     sky_idx=case["sky_idx"])
    return {"flood": flood.numpy()}


def maskspade_shapes():
    """Default generator with gen.m.use_spade: encoder / depth / seg as in the default masker + the SPADE mask decoder."""
    import json

    shapes = {k: v for k, v in masker_shapes().items() if not k.startswith("decoders.m.")}
    extra = json.loads((GOLDEN / "generator_maskspade_shapes.json").read_text())
    shapes.update({k: tuple(v) for k, v in extra.items()})
    return shapes


def maskspade_state_dict(case):
    sd = fill.fill_state_dict(maskspade_shapes(), case["seed"], gain=case["gain"])
    return {k: t(v) for k, v in sd.items()}


def run_oracle_maskspade(name, case):
    sd = maskspade_state_dict(case)
    x = t(case_inputs(name, case)["x"])
    with torch.no_grad():
        z = cpu_ref.resnet101(x, sd, "encoder")
        d, z_depth = cpu_ref.dada_depth_decoder(z, sd, "decoders.d", 160)
        s = cpu_ref.deeplab_v3_decoder(z, sd, "decoders.s", (160, 160), z_depth, use_dada=True)
        cond = cpu_ref.make_m_cond(d, s, x)
        m = torch.sigmoid(cpu_ref.mask_spade_decoder(z, cond, sd, "decoders.m"))
        logits2 = cpu_ref.mask_spade_decoder(z, cond, sd, "decoders.m")
    return {"d": d.numpy(), "s": s.numpy(), "cond": cond.numpy(), "m": m.numpy(), "logits2": logits2.numpy()}


def vgg_state_dict(case):
    return {k: t(v) for k, v in fill.fill_state_dict(cpu_ref.vgg19_shapes(), case["seed"], gain=case["gain"]).items()}


def run_oracle_vgg(name, case):
    sd = vgg_state_dict(case)
    inp = {k: t(v) for k, v in case_inputs(name, case).items()}
    x, m = inp["x"], inp["m"]
    fake = inp["fake"].clone().requires_grad_(True)
    loss = cpu_ref.painter_vgg_term(sd, fake, x, m, case["lambda_vgg"])
    (dfake,) = torch.autograd.grad(loss, fake)
    with torch.no_grad():
        a = cpu_ref.vgg_preprocess((x * (1.0 - m) + fake * m) * m)
        fa, fb = cpu_ref.vgg19_features(a, sd), cpu_ref.vgg19_features(cpu_ref.vgg_preprocess(x * m), sd)
    return {"loss": loss.detach().numpy().reshape(1), "dfake": dfake.numpy(),
            "terms": np.array([(u - v).abs().mean().item() for u, v in zip(fa, fb)], dtype=np.float32),
            "feat_absmean": np.array([u.abs().mean().item() for u in fa], dtype=np.float32),
            "pre_fake": a.numpy()}


def run_oracle_hinge(name, case):
    inp = case_inputs(name, case)
    out = {}
    for tag, real, for_d in (("d_real", True, True), ("d_fake", False, True), ("g", True, False)):
        preds = [t(inp["p%d" % i]).clone().requires_grad_(True) for i in range(len(case["sizes"]))]
        loss = cpu_ref.hinge_loss([[p * 0, p] for p in preds], real, for_d)
        grads = torch.autograd.grad(loss, preds)
        out[tag] = loss.detach().numpy().reshape(1)
        for i, g in enumerate(grads):
            out["%s.grad%d" % (tag, i)] = g.numpy()
    return out


def run_oracle(name, case, dtype=torch.float32):
    """Run oracle.cpu_ref on the seeded inputs of a golden case; same output keys as make_golden."""
    if case["kind"] == "extra_adam":
        return run_oracle_extra_adam(name, case)
    if case["kind"] == "masker":
        return run_oracle_masker(name, case)
    if case["kind"] == "infer":
        return run_oracle_infer(name, case)
    if case["kind"] == "cloudy":
        return run_oracle_cloudy(name, case)
    if case["kind"] == "maskspade":
        return run_oracle_maskspade(name, case)
    if case["kind"] == "dstep_p":
        return run_oracle_dstep(name, case)
    if case["kind"] == "gstep_p":
        return run_oracle_gstep(name, case)
    if case["kind"] == "vgg":
        return run_oracle_vgg(name, case)
    if case["kind"] == "hinge":
        return run_oracle_hinge(name, case)
    sd = case_state_dict(case, dtype)
    inp = {k: t(v).to(dtype) for k, v in case_inputs(name, case).items()}
    out = {}
    k = case["kind"]
    with torch.no_grad():
        if k == "spade":
            out["y"] = cpu_ref.spade(inp["x"], inp["seg"], {"s." + a: b for a, b in sd.items()}, "s").numpy()
        elif k == "resblk":
            sd = {"b." + a: b for a, b in sd.items()}
            out["y"] = cpu_ref.spade_resnet_block(inp["x"], inp["seg"], sd, "b").numpy()
            sd = {a[2:]: b for a, b in sd.items()}
        elif k == "painter":
            H, W = case["H"], case["W"]
            y = cpu_ref.painter_forward(sd, inp["cond"], H // 2 ** case["n_up"], W // 2 ** case["n_up"]).numpy()
            if case["full"]:
                out["y"] = y
            else:
                out.update({"y_" + a: b for a, b in summarize(y).items()})
        elif k == "paint":
            H, W = case["H"], case["W"]
            out["y"] = cpu_ref.paint(sd, inp["m"], inp["x"], H // 2 ** case["n_up"], W // 2 ** case["n_up"]).numpy()
        elif k == "disc_p":
            res = cpu_ref.multiscale_discriminator(inp["x"], sd, case["num_D"], case["n_layers"])
            for i, scale in enumerate(res):
                for j, f in enumerate(scale):
                    out["d%d_%d" % (i, j)] = f.numpy()
        elif k == "disc_fc":
            out["y"] = cpu_ref.fc_discriminator(inp["x"], sd).numpy()
    for key, v in sd.items():
        if key.endswith("weight_u"):
            out["post." + key] = v.numpy().copy()
    return out


def load_golden(name):
    with np.load(GOLDEN / (name + ".npz")) as z:
        return {k: z[k] for k in z.files}
