"""Pin the oracle (oracle.cpu_ref) against the committed golden vectors produced by the REAL reference
(oracle/make_golden.py).  CPU-only; runs everywhere (the GPU box has no /root/reference)."""
import numpy as np
import pytest

from helpers import golden_cases, load_golden, run_oracle

CASES = golden_cases()
# fp32 CPU conv kernels (oneDNN) may pick different blocking for the functional vs module call: 1e-5 abs.
TOL = 2e-5


@pytest.mark.parametrize("name", [n for n in CASES if n not in ("painter_640", "masker_small", "infer_small", "dstep_p", "gstep_p", "gstep_p_aux", "gstep_p_local", "gstep_p_local_aux", "cloudy_small", "maskspade_small", "masker_losses", "mstep", "mstep_spade", "vgg_small", "fire_small")])
def test_oracle_matches_golden(name):
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    assert sorted(gold) == sorted(got)
    for k in gold:
        assert gold[k].shape == got[k].shape, k
        err = np.abs(gold[k].astype(np.float64) - got[k].astype(np.float64)).max()
        assert err <= TOL, "%s/%s: max abs err %.3g" % (name, k, err)


def test_oracle_matches_golden_painter_640():
    """Full-size default Painter (latent 640, spade_n_up 7, 640x640): statistics/crops/pooled map."""
    name = "painter_640"
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    for k in gold:
        err = np.abs(gold[k].astype(np.float64) - got[k].astype(np.float64)).max()
        assert err <= 5e-5, "%s/%s: max abs err %.3g" % (name, k, err)


def test_oracle_matches_golden_masker():
    """ResNet-101 encoder + DADA depth + DeepLab-v3+ seg + mask decoder (eval mode) at 128x160: 150+ conv layers in
    fp32 on two different code paths (module vs functional) -> 1e-4 relative to each output's scale."""
    name = "masker_small"
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    assert sorted(gold) == sorted(got)
    for k in gold:
        scale = max(np.abs(gold[k]).max(), 1e-6)
        err = np.abs(gold[k].astype(np.float64) - got[k].astype(np.float64)).max()
        assert err <= 1e-4 * scale, "%s/%s: max abs err %.3g (scale %.3g)" % (name, k, err, scale)


def test_oracle_matches_golden_infer_all():
    """``Trainer.infer_all`` of the REAL reference (flood event, uint8 conversion, uint8 mask) vs the oracle restatement.
    Float tensors to 1e-4 of their scale; the uint8 flood may differ by one level where a float sits on a truncation
    boundary; the binary mask may flip only where the float mask is within 1e-5 of the threshold."""
    name = "infer_small"
    case = CASES[name]
    gold = load_golden(name)
    got = run_oracle(name, case)
    for k in ("m", "d", "s", "flood"):
        scale = max(np.abs(gold[k]).max(), 1e-6)
        err = np.abs(gold[k].astype(np.float64) - got[k].astype(np.float64)).max()
        assert err <= 1e-4 * scale, "%s/%s: max abs err %.3g (scale %.3g)" % (name, k, err, scale)
    flips = gold["mask_u8"] != got["mask_u8"]
    assert (np.abs(gold["m"] - case["bin_value"])[flips] < 1e-5).all()
    assert flips.mean() < 1e-3
    if not flips.any():
        d8 = np.abs(gold["flood_u8"].astype(np.int32) - got["flood_u8"].astype(np.int32))
        assert d8.max() <= 1 and (d8 > 0).mean() < 5e-3, (d8.max(), (d8 > 0).mean())
    assert 0.2 < (gold["mask_u8"] > 0).mean() < 0.5          # the fixture has a real two-valued mask


def _sibling_weight_scale(gold, k):
    base = k.rsplit(".", 1)[0]
    for leaf in ("weight_bar", "weight"):
        if base + "." + leaf in gold:
            return np.abs(gold[base + "." + leaf]).max()
    return 0.0


def test_oracle_matches_golden_painter_g_step():
    """Painted image, GAN / feature-matching loss terms and the gradient of all trainable Painter tensors (through D,
    the paste, SPADE, spectral norm) vs the oracle under torch autograd.  Two fp32 evaluations of a 30-layer network
    with LeakyReLU / ReLU kinks (module vs functional kernels): 5e-3 of each gradient's scale; a bias in front of an
    instance norm has an exactly-zero gradient, so bias gradients are measured against their layer's weight-gradient
    scale."""
    name = "gstep_p"
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    assert sorted(gold) == sorted(got)
    for k in gold:
        scale = max(np.abs(gold[k]).max(), 1e-12)
        if k.endswith("bias"):
            scale = max(scale, _sibling_weight_scale(gold, k))
        err = np.abs(gold[k].astype(np.float64) - got[k].astype(np.float64)).max()
        tol = 1e-4 if not k.startswith("grad.") else 5e-3
        assert err <= tol * scale + 1e-7, "%s/%s: max abs err %.3g (scale %.3g)" % (name, k, err, scale)


def test_oracle_matches_golden_painter_d_step():
    """Loss and every parameter gradient of the Painter discriminator update (reference modules + GANLoss +
    ``backward()``) vs the oracle's functional restatement under torch autograd: 1e-4 of each gradient's scale."""
    name = "dstep_p"
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    assert sorted(gold) == sorted(got)
    for k in gold:
        scale = max(np.abs(gold[k]).max(), 1e-12)
        err = np.abs(gold[k].astype(np.float64) - got[k].astype(np.float64)).max()
        # biases in front of an instance norm have an exactly-zero gradient (the norm removes the mean): fp32 noise only
        assert err <= 1e-4 * scale + 1e-7, "%s/%s: max abs err %.3g (scale %.3g)" % (name, k, err, scale)
    assert any(k.startswith("grad.") and np.abs(v).max() > 0 for k, v in gold.items())


def test_oracle_smog_matches_golden():
    """cpu_ref.compute_smog on the reference's own depth map vs the smog tensor captured inside the reference's
    Trainer.infer_all (and its uint8 image)."""
    import torch
    from helpers import t
    from oracle import cpu_ref
    from oracle.make_golden import case_inputs

    name = "infer_small"
    gold = load_golden(name)
    x = t(case_inputs(name, CASES[name])["x"])
    smog = cpu_ref.compute_smog(x, t(gold["d"]))
    assert np.abs(smog.numpy() - gold["smog"]).max() <= 1e-5
    d8 = np.abs(cpu_ref.to_uint8_hwc(smog).astype(np.int32) - gold["smog_u8"].astype(np.int32))
    assert d8.max() <= 1 and (d8 > 0).mean() < 5e-3


def test_oracle_paint_cloudy_matches_golden():
    """paint_cloudy of the reference generator (RNG seeded before the call) vs the oracle fed the same lattice angles,
    the reference's own segmentation logits and binary mask."""
    name = "cloudy_small"
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    assert 0.05 < float(gold["sky_fraction"][0]) < 0.6           # the fixture really replaces part of the image
    err = np.abs(got["flood"] - gold["flood"]).max()
    assert err <= 2e-5, err


def test_oracle_matches_golden_mask_spade_decoder():
    """make_m_cond + MaskSpadeDecoder (batch-norm SPADE, spectral_batch projections) of the reference generator."""
    name = "maskspade_small"
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    assert sorted(gold) == sorted(got)
    for k in gold:
        scale = max(np.abs(gold[k]).max(), 1e-6)
        err = np.abs(gold[k].astype(np.float64) - got[k].astype(np.float64)).max()
        assert err <= 1e-4 * scale, "%s/%s: max abs err %.3g (scale %.3g)" % (name, k, err, scale)


def test_masker_loss_restatements_match_reference_golden():
    """oracle.cpu_ref's loss restatements against the values and input gradients the reference's own loss classes
    produced (fixture masker_losses, oracle/make_golden.py::run_reference_masker_losses)."""
    import torch
    from helpers import t
    from oracle import cpu_ref
    from oracle.make_golden import case_inputs

    name = "masker_losses"
    case = golden_cases()[name]
    gold = load_golden(name)
    inp = {k: t(v) for k, v in case_inputs(name, case).items()}

    def check(key, fn, leaf_name, half=False):
        leaf = inp[leaf_name].clone()
        if half:
            leaf = leaf.half().float()
        leaf.requires_grad_(True)
        loss = fn(leaf)
        assert abs(loss.item() - float(gold[key][0])) <= 1e-6 * max(1.0, abs(float(gold[key][0]))), key
        if key + ".grad" in gold:
            (g,) = torch.autograd.grad(loss, leaf)
            ref = gold[key + ".grad"]
            assert np.abs(g.numpy() - ref).max() <= 1e-6 * max(np.abs(ref).max(), 1e-12) + 1e-9, key

    check("crossent", lambda s: cpu_ref.cross_entropy(s, inp["s_target"]), "s_logits")
    check("minent_v1", lambda s: cpu_ref.minent_loss(torch.softmax(s, dim=1)), "s_logits")
    check("entropy_dada_sum", lambda s: (cpu_ref.prob_2_entropy(torch.softmax(s, dim=1)) * inp["d_pred"] * 0.37).sum(),
          "s_logits")
    check("bce", lambda m: torch.nn.functional.binary_cross_entropy_with_logits(m, inp["m_target"]), "m_logits")
    check("tv", lambda m: cpu_ref.tv_loss(torch.sigmoid(m)), "m_logits")
    check("minent_v2", lambda m: cpu_ref.minent_loss(torch.cat([torch.sigmoid(m), 1 - torch.sigmoid(m)], dim=1), 2, 0.1),
          "m_logits")
    assert cpu_ref.ground_intersection_loss(torch.sigmoid(inp["m_logits"]), inp["ground"]).item() == float(gold["gi"][0])
    check("advent_wgan_0", lambda d: cpu_ref.advent_wgan(d, 0), "d_out")
    check("advent_wgan_1", lambda d: cpu_ref.advent_wgan(d, 1), "d_out")
    check("sigm", lambda d: cpu_ref.sigm_loss(d, inp["depth_target"]), "depth_pred", half=True)


def test_oracle_vgg_term_matches_reference_golden():
    """A18: ``cpu_ref.vgg_preprocess`` / ``vgg19_features`` / ``vgg_loss`` vs the reference's own ``vgg_preprocess``,
    ``Vgg19`` and ``VGGLoss`` (golden ``vgg_small``): pre-processed image, the five per-tap L1 terms, the weighted loss and
    its gradient w.r.t. the painter output."""
    name = "vgg_small"
    gold = load_golden(name)
    got = run_oracle(name, CASES[name])
    assert sorted(gold) == sorted(got)
    np.testing.assert_allclose(got["pre_fake"], gold["pre_fake"], rtol=0, atol=2e-5)
    for k in ("loss", "terms", "feat_absmean"):
        np.testing.assert_allclose(got[k], gold[k], rtol=2e-5, atol=0)
    scale = np.abs(gold["dfake"]).max()
    assert np.abs(got["dfake"] - gold["dfake"]).max() <= 1e-4 * scale


def test_oracle_add_fire_matches_reference_fire_py():
    """N1 wildfire: ``cpu_ref.add_fire`` vs the reference's OWN ``fire.add_fire`` (golden ``fire_small``; only kornia's
    Gaussian / filter2d and torchvision's adjust_contrast / adjust_brightness were bound to their documented formulas,
    the blur there as a direct fp64 2-D correlation).  Byte image: identical except where the separable fp32 blur lands
    a paste within rounding of an integer boundary -- at most one level on < 1e-3 of the values."""
    import torch
    from helpers import case_inputs, t
    from oracle import cpu_ref

    name = "fire_small"
    case, gold = CASES[name], load_golden(name)
    inp = {k: t(v) for k, v in case_inputs(name, case).items()}
    y = cpu_ref.add_fire(inp["x"], inp["seg"], float(gold["green"][0]), sky_idx=case["sky_idx"]).numpy()
    d = np.abs(y - gold["y_u8"].astype(np.float32))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3, (d.max(), (d > 0).mean())
    assert 100 <= int(gold["green"][0]) <= 150
