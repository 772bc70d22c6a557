"""BASELINE configs[2], [3] and [4] at FULL SIZE (640 x 640, the benchmark batch sizes) against goldens produced by the
reference's own ``Trainer`` at 640 x 640 (oracle/make_golden_640.py):

* ``infer_640``  <- ``Trainer.infer_all`` (2 images)            -> configs[4]: ``infer_all`` bs 16 fp16, 3 events
* ``jstep_640``  <- ``Trainer.update_G`` + ``Trainer.update_D``   -> configs[3] per-GPU slice: joint step, 4 per domain;
                    (2 samples per domain, all default tasks)      configs[2]: Masker train step, bs 8

The benchmark batches are the golden batch REPEATED (x8, x2, x4).  For inference samples are independent, so every
repeat must reproduce the golden images.  For training, repeating a batch leaves every batch statistic (BatchNorm mean
/ biased variance, SIGM's batch median), every mean-reduced loss term and therefore every parameter gradient unchanged
(the one sum-reduced term, SIGM, is re-weighted through its two config lambdas, see _build_train)
-- a bs-8 step on 4 x the golden batch has the golden step's loss terms and gradients, which exercises the kernel
selections that only appear at these sizes (cooperative weight-gradient tiles, 128 x 256 GEMM tiles, one-chunk 3x3 ...).

Weights: the well-conditioned portable fill (bottleneck bn3 gamma ~ 0.05), for which the reference's OWN gradients keep
their direction under 16-bit storage (tests/devtools/measure_ref_grad_quant2.py), so directions are asserted."""
import random

import numpy as np
import pytest
import torch

from helpers import load_golden, t
from oracle.make_golden import summarize
from oracle.make_golden_640 import CASES_640, generator_fill, infer_inputs, jstep_inputs

pytestmark = pytest.mark.gpu


def _load(mod, sd_np):
    mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()}, strict=True)


# ------------------------------------------------------------------------------------------------ configs[4]
# (max, mean) deviation of the reference's OWN 16-bit run (G.half() / G.bfloat16() on the CPU) from its fp32 run on this
# fixture, per stage output, relative to the output's max |value| (tests/devtools/measure_ref_half_masker.py)
REF_HALF_MASKER = {
    ("d", "bfloat16"): (0.03825, 0.007637),
    ("s", "bfloat16"): (0.02003, 0.003674),
    ("m", "bfloat16"): (0.08916, 0.002878),
    ("d", "float16"): (0.004552, 0.0009773),
    ("s", "float16"): (0.003698, 0.000626),
    ("m", "float16"): (0.007331, 0.0003074),
}


@pytest.fixture(scope="module")
def infer_trainer():
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    case = CASES_640["infer_640"]
    opts = default_opts()
    opts.tasks = ["d", "s", "m", "p"]
    T = Trainer(opts, device="cuda").setup(inference=True)
    shapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
    sd = generator_fill(shapes, case)
    _load(T.G, sd)
    T.G.set_compute_dtype(torch.float16)
    return T, sd, case


def test_apply_events_bs16_fp16_matches_reference_infer_all(infer_trainer):
    T, sd, case = infer_trainer
    gold = load_golden("infer_640")
    B, H, W = case["B"], case["H"], case["W"]
    x2 = t(infer_inputs(case)["x"]).cuda()
    x = x2.repeat(8, 1, 1, 1)                                      # bs 16 = BASELINE configs[4]
    _load(T.G, sd)
    random.seed(case["rng_seed"])                                  # fire.py:115 draws the filter's green level
    out = T.infer_all(x, numpy=True, bin_value=case["bin_value"], half=True, return_masks=True)
    assert set(out) == {"flood", "wildfire", "smog", "mask"}
    for k in ("flood", "wildfire", "smog"):
        assert out[k].shape == (16, H, W, 3) and out[k].dtype == np.uint8
    assert out["mask"].shape == (16, 1, H, W) and set(np.unique(out["mask"])) <= {0, 255}

    # --- the binary flood mask: bit-exact outside the fp16 noise band of the threshold (|m - 0.5| < 0.11 of the reference's
    # float mask = |logit| < 0.45: the reference's own G.half() run moves the logit by up to 0.34 on this fixture)
    ref_mask = np.unpackbits(gold["mask_bits"])[: B * H * W].reshape(B, 1, H, W).astype(bool)
    band = np.unpackbits(gold["m_band"])[: B * H * W].reshape(B, 1, H, W).astype(bool)
    assert band.mean() < 0.05                                      # >= 95 % of the pixels are decided away from 0.5
    got_mask = out["mask"] > 0
    for i in range(16):
        r, b = ref_mask[i % B], band[i % B]
        assert np.array_equal(got_mask[i][~b], r[~b]), "sample %d: mask differs outside the threshold band" % i
        agree = (got_mask[i] == r).mean()
        assert agree >= 0.99, (i, agree)
    print("\nmask: band %.3g of the pixels, agreement inside it %.4f"
          % (band.mean(), (got_mask[:B] == ref_mask)[band].mean()))

    # --- independence of the samples: every repeat of an image gives the same bytes
    for k in ("flood", "wildfire", "smog", "mask"):
        for i in range(B, 16):
            if not np.array_equal(out[k][i], out[k][i % B]):
                d = np.argwhere(out[k][i] != out[k][i % B])
                raise AssertionError("%s: sample %d differs from its original %d in %d values, index ranges %s .. %s, largest "
                                     "difference %d" % (k, i, i % B, len(d), d.min(0).tolist(), d.max(0).tolist(),
                                                        np.abs(out[k][i].astype(int) - out[k][i % B].astype(int)).max()))

    # --- the three uint8 events vs the reference's own uint8 images (crops, 8x pooled map, per-channel statistics)
    report = {}
    for k in ("flood", "wildfire", "smog"):
        u8 = np.ascontiguousarray(out[k][:B].transpose(0, 3, 1, 2)).astype(np.float32)
        s = summarize(u8)
        crops = np.concatenate([np.abs(s[c] - gold["%s_u8_%s" % (k, c)]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
        pooled = np.abs(s["pooled8"] - gold[k + "_u8_pooled8"])
        report[k] = (crops.max(), crops.mean(), (crops > 1).mean(), pooled.max(), np.abs(s["mean"] - gold[k + "_u8_mean"]).max())
        print("%s u8 vs reference: crops max %g mean %.3g, >1 level on %.3g; pooled8 max %.3g; channel mean %.3g"
              % ((k,) + report[k]))
    # smog is byte-image arithmetic on x and the depth map: levels agree except where a 16-bit depth difference crosses a
    # rounding boundary.  The wildfire's only input from the network is the segmentation ARG-MAX (sky = class 9, fire.py):
    # like the flood mask it is a decision, and where the untrained logits nearly tie (top-2 margin < 0.1 on 5.9 % of the
    # pixels of this fixture) a 16-bit activation chain decides differently -- the reference's own fp16 run does.  So, as
    # for the mask: (1) the arg-max equals the reference's OUTSIDE its near-tie band, (2) the event on the REFERENCE's
    # arg-max reproduces the reference's bytes (the wildfire kernels themselves), (3) end to end only a loose bound (a
    # flipped sky pixel moves the image-wide mean of adjust_contrast and, dilated and blurred, its neighbourhood).
    # (Until round 3 (2) was asserted end to end, which pinned the fp16 path to the plain GEMM kernel's fp32 summation
    # order: the dispatch was keyed on the dtype for this fixture's sake.  It is shape-only now.)
    with torch.no_grad():
        zz = T.G.encode(x2.half())
        _, zd = T.G.decoders["d"].forward_nhwc(zz)
        seg_hip = T.G.decoders["s"].forward_nhwc(zz, zd)
    am_hip = seg_hip.t[..., :11].float().argmax(-1).cpu().numpy()
    am_ref = gold["seg_argmax"]
    tie = np.unpackbits(gold["seg_tie_band"])[: am_ref.size].reshape(am_ref.shape).astype(bool)
    assert float(gold["seg_tie_frac"][0]) < 0.08
    assert np.array_equal(am_hip[~tie], am_ref[~tie]), "segmentation arg-max differs outside the near-tie band"
    assert (am_hip == am_ref).mean() >= 0.985, (am_hip == am_ref).mean()
    print("seg arg-max: agreement %.4f overall, %.4f inside the near-tie band (%.3g of the pixels)"
          % ((am_hip == am_ref).mean(), (am_hip == am_ref)[tie].mean(), tie.mean()))
    from climategan_amd import ops
    onehot = torch.zeros((B, am_ref.shape[1], am_ref.shape[2], 16), dtype=torch.float16, device="cuda")
    onehot.scatter_(3, torch.from_numpy(am_ref.astype(np.int64)).cuda()[..., None], 10.0)
    f = T.opts.events.fire
    wf = ops.wildfire(x2.half(), ops.NHWC(onehot, 11), float(gold["green"][0]), kernel_size=f.get("kernel_size", 301),
                      kernel_sigma=f.get("kernel_sigma", 150.5), transparency=200, crop_bottom=bool(f.get("crop_bottom_sky_mask")))
    wf_u8 = ops.normalize_to_uint8(wf.to(torch.float16)).cpu().numpy().transpose(0, 3, 1, 2)      # as infer_all does
    sw = summarize(np.ascontiguousarray(wf_u8).astype(np.float32))
    wcrops = np.concatenate([np.abs(sw[c] - gold["wildfire_u8_%s" % c]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
    wrep = (wcrops.max(), wcrops.mean(), (wcrops > 1).mean(), np.abs(sw["pooled8"] - gold["wildfire_u8_pooled8"]).max(),
            np.abs(sw["mean"] - gold["wildfire_u8_mean"]).max())
    print("wildfire on the reference's arg-max, u8 vs reference: crops max %g mean %.3g, >1 level on %.3g; pooled8 max %.3g; "
          "channel mean %.3g" % wrep)
    assert wrep[2] < 1e-2 and wrep[4] < 0.1, wrep
    assert report["wildfire"][4] < 2.0 and report["wildfire"][3] < 16.0, report["wildfire"]
    assert report["smog"][1] < 0.75 and report["smog"][3] < 2.0 and report["smog"][4] < 0.5, report["smog"]
    # end to end the flood also carries the mask bits that flipped inside the band (each flips a pixel between "painted"
    # and "original": up to 255 levels): only its channel means are compared here, the painter itself below
    assert report["flood"][4] < 0.75, report["flood"]

    # --- the flood painter on the REFERENCE's binary mask (no flipped bits): float image vs the reference's float flood
    # (crops + 8x pooled map), bounded by the painter's own 16-bit yardstick (tests/test_gpu_painter.py, painter_640 fp16:
    # reference .half() deviates max 1.05e-2, mean 1.29e-3 from its fp32 run) x 1.25
    _load(T.G, sd)
    m_ref = torch.from_numpy(ref_mask.astype(np.float32)).cuda()
    T.G.painter.set_latent_shape((B, 3, H, W), True)
    with torch.no_grad():
        flood = T.compute_flood(x2.half(), m=m_ref.half(), bin_value=case["bin_value"]).float().cpu().numpy()
    s = summarize(flood)
    err = np.concatenate([np.abs(s[c] - gold["flood_" + c]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
    pooled = np.abs(s["pooled8"] - gold["flood_pooled8"]).max()
    print("flood painter on the reference mask: crops max %.3g mean %.3g, pooled8 max %.3g" % (err.max(), err.mean(), pooled))
    assert np.percentile(err, 99.9) <= 1.25 * 1.05e-2 and err.mean() <= 1.25 * 1.29e-3 and pooled <= 1.25 * 1.05e-2
    outside = ~ref_mask.repeat(3, axis=1)
    assert np.array_equal(flood[outside], x2.half().float().cpu().numpy()[outside])      # paste: original pixels, bit-exact


def test_masker_stages_640_vs_reference(infer_trainer):
    """Depth / segmentation / mask float outputs of the Masker at 640 x 640 (fp16 and bf16) vs the reference's fp32 run,
    bounded by what the reference's own 16-bit run does on this fixture (x 1.25)."""
    T, sd, case = infer_trainer
    gold = load_golden("infer_640")
    x = t(infer_inputs(case)["x"]).cuda()
    for dt in (torch.float16, torch.bfloat16):
        _load(T.G, sd)
        T.G.set_compute_dtype(dt)
        with torch.no_grad():
            out = T.G.masker_forward(x)
        for k in ("d", "s", "m"):
            y = out[k].float().cpu().numpy()
            s = summarize(y)
            scale = max(np.abs(gold[k + "_crop_c"]).max(), np.abs(gold[k + "_pooled8"]).max())
            errs = np.concatenate([np.abs(s[c] - gold["%s_%s" % (k, c)]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
            if k == "m":
                # saturated sigmoid: compare away from the logit's zero crossings
                ref = np.concatenate([gold["m_" + c].ravel() for c in ("crop_tl", "crop_c", "crop_br")])
                errs = errs[np.abs(ref - 0.5) > 0.45]
            name = str(dt).split(".")[1]
            print("\nmasker 640 %s %s: max %.3g mean %.3g of scale %.3g" % (k, name, errs.max(), errs.mean(), scale))
            bound = REF_HALF_MASKER.get((k, name))
            if bound is not None:
                assert errs.max() <= 1.25 * bound[0] * scale and errs.mean() <= 1.25 * bound[1] * scale, (k, name)
            else:
                assert errs.max() <= (3e-2 if dt == torch.float16 else 0.2) * scale, (k, name)
    T.G.set_compute_dtype(torch.float16)


@pytest.mark.parametrize("mode", ["split24", "pair16"])
def test_split_precision_masker_reproduces_the_fp32_reference(infer_trainer, mode):
    """``G.float()`` on the eval-mode generator = the split-precision Masker ("split24": every activation as three bf16
    numbers hi + mid + lo; "pair16": two fp16 numbers; csrc/pair.hip) through the same MFMA kernels: the arithmetic of the
    reference's DEFAULT apply_events run (fp32; --half is opt-in, apply_events.py:465-468).  Against the reference's own
    fp32 ``infer_all`` at 640 x 640, bs 16:
    * depth / segmentation / mask floats within 2e-5 of their scale (north_star asks 1e-3): 6e-6 / 4.5e-6 / 3.5e-5 measured,
      which is what the reference's fp32 run itself is away from float64 on this fixture (2.6e-6 / 4.1e-6 / 2.9e-5,
      tests/devtools/measure_ref_fp32_floor.py) -- the noise floor of ANY fp32 evaluation;
    * the binarised flood mask ("bit-exact", trainer.py:1866-1871) equal to the reference's on every pixel fp32 arithmetic
      can decide: the fixture stores the 69 pixels (of 819 200) whose FLOAT64 logit is inside 8 x the reference's own fp32
      rounding noise (1.3e-4) of the threshold; outside them no bit may differ, inside at most 2 per image pair (measured:
      1, at a logit of -2.7e-5, where MKL-DNN's and the MFMA tiles' fp32 summation orders disagree).  No fp16 band any more
      (test_apply_events_bs16: 3.8 % of the pixels in fp16)."""
    T, sd, case = infer_trainer
    gold = load_golden("infer_640")
    B, H, W = case["B"], case["H"], case["W"]
    x2 = t(infer_inputs(case)["x"]).cuda()
    _load(T.G, sd)
    T.G.eval()
    if mode == "split24":
        assert T.G.float() is T.G and T.G.pair_precision and T.G.compute_dtype == torch.bfloat16
    else:
        T.G.set_compute_dtype("pair16")
    try:
        with torch.no_grad():
            out = T.G.masker_forward(x2)
        worst = {}
        for k in ("d", "s", "m"):
            y = out[k].float().cpu().numpy()
            assert out[k].dtype == torch.float32
            s = summarize(y)
            scale = max(np.abs(gold[k + "_crop_c"]).max(), np.abs(gold[k + "_pooled8"]).max())
            errs = np.concatenate([np.abs(s[c] - gold["%s_%s" % (k, c)]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
            worst[k] = errs.max() / scale
            print("%s masker 640 %s: max %.3g mean %.3g of scale %.3g (%.2g relative)" % (mode, k, errs.max(), errs.mean(), scale, worst[k]))
            assert errs.max() <= 1e-3 * scale, (k, errs.max(), scale)          # north_star's 1e-3 ...
            assert errs.max() <= (1e-4 if k == "m" else 2e-5) * scale, (k, errs.max(), scale)   # ... and what fp32-grade arithmetic gives
            # (m: the fixture's output conv has gain 40, the sigmoid's slope is 1/4)
        _load(T.G, sd)                                                        # (spectral-norm u / v back to the fixture's)
        random.seed(case["rng_seed"])
        res = T.infer_all(x2.repeat(8, 1, 1, 1), numpy=True, bin_value=case["bin_value"], half=False, return_masks=True)
        ref_mask = np.unpackbits(gold["mask_bits"])[: B * H * W].reshape(B, 1, H, W).astype(bool)
        got = res["mask"] > 0
        diff = sum(int((got[i] != ref_mask[i % B]).sum()) for i in range(16))
        print("%s flood mask vs the reference's fp32 mask: %d of %d bits differ" % (mode, diff, got.size))
        band = np.unpackbits(gold["m_fp32_band"])[: B * H * W].reshape(B, 1, H, W).astype(bool)
        assert int(gold["m_fp32_band_count"][0]) == band.sum() <= 100 and int(gold["m_fp32_vs_fp64_flips"][0]) <= 2
        for i in range(16):
            d = got[i] != ref_mask[i % B]
            assert not d[~band[i % B]].any(), "sample %d: %d mask bits differ where fp32 decides" % (i, d[~band[i % B]].sum())
            assert np.array_equal(got[i], got[i % B])                          # every repeat of an image: the same bits
        assert diff // 8 <= 2, diff
        for k in ("flood", "wildfire", "smog"):
            assert res[k].shape == (16, H, W, 3) and res[k].dtype == np.uint8
        # round 5: the Painter runs on split maps as well (G.painter.pair_precision), so the flood IMAGE is the fp32 reference's
        # up to uint8 truncation boundaries and the (at most one) mask pixel inside the fp32 band: levels within 1 on >= 99.9 %
        # of the crop pixels (measured: one level on 5e-5 of them in split24, 1.6e-4 in pair16; round 4's all-gather-kernel build: identical bytes in split24), channel means within
        # 0.05 of a level (the 16-bit Painter: channel means only, 0.75)
        assert T.G.painter.pair_precision
        u8 = np.ascontiguousarray(res["flood"][:B].transpose(0, 3, 1, 2)).astype(np.float32)
        sf = summarize(u8)
        crops = np.concatenate([np.abs(sf[c] - gold["flood_u8_%s" % c]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
        print("%s flood u8 vs the reference's fp32 run: crops max %g mean %.3g, > 1 level on %.3g; pooled8 max %.3g; channel "
              "mean %.3g" % (mode, crops.max(), crops.mean(), (crops > 1).mean(), np.abs(sf["pooled8"] - gold["flood_u8_pooled8"]).max(),
                             np.abs(sf["mean"] - gold["flood_u8_mean"]).max()))
        assert (crops > 1).mean() <= 1e-3 and crops.mean() <= 0.3, (crops.max(), crops.mean(), (crops > 1).mean())
        # (8 x 8 pooled map: the one mask pixel inside the fp32 band flips a pixel between "painted" and "original", up to
        # 255 / 64 = 4 levels of one pooled cell; measured 1.38.  Everywhere else the bytes are the reference's: crops max 0)
        assert np.abs(sf["pooled8"] - gold["flood_u8_pooled8"]).max() <= 4.0
        assert np.abs(sf["mean"] - gold["flood_u8_mean"]).max() <= 0.05
    finally:
        T.G.set_compute_dtype(torch.float16)
        _load(T.G, sd)


@pytest.mark.parametrize("mode", ["split24", "pair16"])
def test_hybrid_inference_keeps_the_fp32_grade_mask(infer_trainer, mode):
    """Round 6, ``G.set_painter_compute_dtype``: the split-precision Masker with the Painter back on 16 bit.  The flood mask --
    the output north_star wants bit-exact -- is the Masker's alone: it must be the reference's fp32 mask on every pixel fp32
    arithmetic can decide, exactly like the full split-precision run (same fixture, same band as
    test_split_precision_masker_reproduces_the_fp32_reference), and identical to that run's mask bit for bit; the flood IMAGE
    then carries the 16-bit Painter's tolerance (channel means within 0.75 of a uint8 level, the bound of the fp16 run)."""
    T, sd, case = infer_trainer
    gold = load_golden("infer_640")
    B, H, W = case["B"], case["H"], case["W"]
    x2 = t(infer_inputs(case)["x"]).cuda()
    try:
        masks, floods = {}, {}
        for hybrid in (False, True):
            _load(T.G, sd)
            T.G.eval()
            T.G.set_compute_dtype(mode)
            if hybrid:
                assert T.G.set_painter_compute_dtype(torch.float16) is T.G
                assert T.G.pair_precision and T.G.encoder.pair_precision and not T.G.painter.pair_precision
                assert T.G.painter.compute_dtype == torch.float16
            random.seed(case["rng_seed"])
            res = T.infer_all(x2.repeat(8, 1, 1, 1), numpy=True, bin_value=case["bin_value"], half=False, return_masks=True)
            masks[hybrid], floods[hybrid] = res["mask"] > 0, res["flood"]
            for k in ("flood", "wildfire", "smog"):
                assert res[k].shape == (16, H, W, 3) and res[k].dtype == np.uint8
        assert np.array_equal(masks[True], masks[False])                  # the Masker does not know what the Painter runs in
        ref_mask = np.unpackbits(gold["mask_bits"])[: B * H * W].reshape(B, 1, H, W).astype(bool)
        band = np.unpackbits(gold["m_fp32_band"])[: B * H * W].reshape(B, 1, H, W).astype(bool)
        diff = 0
        for i in range(16):
            d = masks[True][i] != ref_mask[i % B]
            assert not d[~band[i % B]].any(), "sample %d: %d mask bits differ where fp32 decides" % (i, d[~band[i % B]].sum())
            diff += int(d.sum())
        assert diff // 8 <= 2, diff
        # the painted image: the 16-bit Painter against the reference's fp32 run (uint8 levels)
        u8 = np.ascontiguousarray(floods[True][:B].transpose(0, 3, 1, 2)).astype(np.float32)
        sf = summarize(u8)
        crops = np.concatenate([np.abs(sf[c] - gold["flood_u8_%s" % c]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
        print("hybrid %s flood u8 vs the reference's fp32 run: crops max %g mean %.3g; channel mean %.3g"
              % (mode, crops.max(), crops.mean(), np.abs(sf["mean"] - gold["flood_u8_mean"]).max()))
        assert np.abs(sf["mean"] - gold["flood_u8_mean"]).max() <= 0.75 and crops.mean() <= 1.0
        # outside the mask the paste keeps the original pixels: identical bytes in both runs
        keep = ~masks[True].repeat(3, axis=1).transpose(0, 2, 3, 1)
        assert np.array_equal(floods[True][keep], floods[False][keep])
    finally:
        T.G.set_compute_dtype(torch.float16)
        _load(T.G, sd)


# ------------------------------------------------------------------------------------------------ configs[2], [3]
def _build_train(tasks, case, reps, dt=torch.bfloat16, merge=True, opt_overrides=None):
    from climategan_amd import fill
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    opts = default_opts()
    opts.tasks = list(tasks)
    opts.dis.soft_shift = 0.0
    opts.dis.flip_prob = 0.0
    # SIGMLoss is the one term that is NOT a batch mean (losses.py:237-278): its data term is summed over the batch (x reps
    # on a repeated batch) and its Sobel term carries the reference's B-fold filter quirk on top (x reps^2).  With
    # lambdas.G.d.main / reps and gml / reps the repeated batch reproduces the golden step's depth term AND gradients.
    opts.train.lambdas.G.d.main = 1.0 / reps
    opts.train.lambdas.G.d.gml = 0.5 / reps
    if "latent_dim" in case:             # the small configuration (jstep_small)
        opts.gen.p.latent_dim, opts.gen.p.spade_n_up = case["latent_dim"], case["n_up"]
        opts.dis.p.ndf, opts.dis.p.n_layers = case["ndf"], case["n_layers"]
    if (opt_overrides or {}).get("m_use_dada"):
        opts.gen.m.use_dada = True
    T = Trainer(opts, device="cuda").setup(inference=False)
    T.merge_domains = merge
    if (case["H"], case["W"]) != (640, 640):
        T.G.decoders["d"]._target_size = case["W"] // 4
        T.G.decoders["s"].set_target_size((case["H"] // 4, case["W"] // 4))
    gshapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
    _load(T.G, generator_fill(gshapes, case))
    dshapes = {k: tuple(v.shape) for k, v in T.D.state_dict().items()}
    _load(T.D, fill.fill_state_dict(dshapes, case["seed"] + 1))
    if "p" in tasks:
        from oracle import cpu_ref
        vgg = T.losses["G"]["p"]["vgg"].vgg
        _load(vgg, fill.fill_state_dict(cpu_ref.vgg19_shapes(), case["vgg_seed"], gain=case["vgg_gain"]))
    T.G.set_compute_dtype(dt)
    T.D.set_compute_dtype(dt)
    return T


def _batch(case, reps, domains):
    inp = jstep_inputs(case)
    out = {}
    for dom in domains:
        out[dom] = {"data": {k: t(v).cuda().repeat(reps, *([1] * (v.ndim - 1))) for k, v in inp[dom].items()}}
    return out


def _grad_sub(key, g, n):
    from climategan_amd import fill
    flat = g.reshape(-1)
    if flat.numel() <= n:
        return flat.float().cpu().numpy()
    idx = (fill.uniform01((n,), fill.key_seed(key, 4242)) * flat.numel()).astype(np.int64).clip(0, flat.numel() - 1)
    return flat[torch.from_numpy(idx).to(flat.device)].float().cpu().numpy()


def _compare_grads(module, prefix, gold, sub, report):
    """Per trainable tensor: gradient norm ratio and the cosine on the golden's seeded sub-sample.  Tensors whose TRUE
    gradient is zero -- a conv bias in front of a BatchNorm / instance norm: the reference's own value is fp32 noise, 1e-4
    of the sibling weight's gradient or less -- are only checked to be small here too."""
    rows = []
    for key, p in module.named_parameters():
        gk = "gnorm.%s.%s" % (prefix, key)
        if gk not in gold:
            continue
        if p.grad is None and not p.requires_grad and key.endswith(("weight_u", "weight_v")):
            # a reference quirk this package does not copy: painter_loss_for_masker ends by setting requires_grad = True on
            # EVERY Painter parameter (trainer.py:1647-1649), the spectral-norm u / v vectors included, so from the first
            # pl4m call on the reference differentiates sigma w.r.t. u and v and lets the optimizer step them.  Here the
            # power-iteration state stays what norms.py:129-133 declares it to be: not trainable.
            continue
        assert p.grad is not None, key
        ref_n = float(gold[gk][0])
        got_n = float(p.grad.norm())
        base = key.rsplit(".", 1)[0]
        sib = max([float(gold[n][0]) for n in ("gnorm.%s.%s.weight" % (prefix, base), "gnorm.%s.%s.weight_bar" % (prefix, base))
                   if n in gold] + [0.0])
        if ref_n < 1e-4 * max(sib, 1e-2):
            assert got_n <= 2e-2 * max(sib, 1e-2), (key, got_n, sib)       # 16-bit noise instead of fp32 noise
            continue
        a, b = gold["gsub.%s.%s" % (prefix, key)].astype(np.float64), _grad_sub(key, p.grad, sub).astype(np.float64)
        cos = float((a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
        rows.append((key, ref_n, got_n / max(ref_n, 1e-30), cos, p.numel()))
    report.extend(rows)
    return rows


def _summ(rows, sel):
    r = [x for x in rows if sel(x[0])]
    ratios = np.array([x[2] for x in r])
    cos = np.array([x[3] for x in r])
    return len(r), np.median(ratios), ratios.min(), ratios.max(), np.median(cos), np.percentile(cos, 10), cos.min()


MASKER_LOSS_KEYS = {  # golden key (reference logger.losses.gen: task.loss.domain) -> Trainer.loss_log key
    "G.task.s.crossent.s": "G.s.crossent.s", "G.task.s.minent.r": "G.s.minent.r", "G.task.s.advent.r": "G.s.advent.r",
    "G.task.m.tv.r": "G.m.tv.r", "G.task.m.tv.s": "G.m.tv.s", "G.task.m.bce.s": "G.m.bce.s", "G.task.m.gi.r": "G.m.gi.r",
    "G.task.m.minent.r": "G.m.minent.r", "G.task.m.advent.r": "G.m.advent.r", "G.task.d.s": "G.d.s",
}


def _check_terms(T, gold, mapping, rel, what):
    for gk, hk in mapping.items():
        if gk not in gold:
            continue
        ref, got = float(gold[gk][0]), float(T.loss_log[hk])
        print("  %-24s reference %+.6g   hip %+.6g" % (hk, ref, got))
        if hk == "G.m.gi.r":        # GroundIntersection counts pixels across a 0.5 threshold: discontinuous in the mask
            assert abs(got - ref) <= 0.25 * abs(ref) + 1e-4, (what, hk, got, ref)
            continue
        assert abs(got - ref) <= rel * max(abs(ref), 1e-3), (what, hk, got, ref)


@pytest.mark.parametrize("config", ["configs2_masker_bs8", "configs3_joint_4_per_domain", "configs3_joint_32_per_domain",
                                    "small_joint_merged", "small_joint_per_domain"])
def test_train_step_640_matches_reference_update(config):
    """One ``Trainer.train_step`` (update_G + update_D) at the benchmark batch size, bf16, vs the reference's own
    ``update_G`` / ``update_D`` at 640 x 640 (golden ``jstep_640``): logged loss terms, per-tensor gradient norms and
    directions for every trainable G and D tensor, BatchNorm running statistics.  The two ``small_*`` configurations run
    the 128 x 160 fixture ``jstep_small`` with the real and the simulated domain going through the Masker trunk as ONE
    batch (grouped BatchNorm, the default) and one after the other (``merge_domains = False``, the reference's call
    sequence): both must reproduce the reference's step."""
    name = "jstep_small" if config.startswith("small") else "jstep_640"
    case = CASES_640[name]
    gold = load_golden(name)
    merge = config != "small_joint_per_domain"
    if config == "configs2_masker_bs8":
        tasks, reps, domains = ("d", "s", "m"), 4, ("r", "s")
    elif config == "configs3_joint_4_per_domain":
        tasks, reps, domains = ("d", "s", "m", "p"), 2, ("r", "s", "rf")
    elif config == "configs3_joint_32_per_domain":
        # BASELINE configs[3] at its GLOBAL batch on one GPU (bench.py's N = 1 headline): the golden pair repeated 16 x;
        # every loss term is a batch mean (SIGMLoss rescaled in _build_train), BatchNorm statistics of a repeated batch are
        # those of one copy, so the reference's step on the pair is the reference's step on the 32.  Maps of 2 GiB and
        # more go through the batch-chunked path of ops.py here and nowhere else in the suite
        tasks, reps, domains = ("d", "s", "m", "p"), 16, ("r", "s", "rf")
    else:
        tasks, reps, domains = ("d", "s", "m", "p"), 1, ("r", "s", "rf")
    T = _build_train(tasks, case, reps, merge=merge)
    batch = _batch(case, reps, domains)
    if "p" in tasks:
        T.G.painter.set_latent_shape((case["B"] * reps, 3, case["H"], case["W"]), True)
    g_loss = T.update_G(batch)
    assert torch.isfinite(g_loss)
    assert_g_side(T, gold, case, name, config, tasks)
    d_loss = T.update_D(batch)
    assert torch.isfinite(d_loss)
    assert_d_side(T, gold, case, name, config, tasks)


def assert_g_side(T, gold, case, name, config, tasks):
    """After ``update_G``: logged loss terms, per-tensor gradient norms / directions of G, BatchNorm running statistics."""
    print("\n%s: G-side loss terms" % config)
    _check_terms(T, gold, MASKER_LOSS_KEYS, 3e-2, config)
    if "p" in tasks:
        _check_terms(T, gold, {"G.p.vgg": "G.p.vgg", "G.p.gan": "G.p.gan", "G.p.featmatch": "G.p.featmatch"}, 2e-2, config)
    rows = []
    _compare_grads(T.G, "G", gold, case["sub"], rows)
    is_conv = lambda k: k.endswith("weight_bar") or (k.endswith(".weight") and ".bn" not in k and ".norm" not in k)
    groups = [("encoder conv", lambda k: k.startswith("encoder.") and is_conv(k)),
              ("encoder bn", lambda k: k.startswith("encoder.") and not is_conv(k)),
              ("decoders", lambda k: k.startswith("decoders."))]
    if "p" in tasks:
        groups.append(("painter", lambda k: k.startswith("painter.")))
    stats = {}
    for gname, sel in groups:
        stats[gname] = _summ(rows, sel)
        if gname == "decoders":
            for r in sorted((x for x in rows if sel(x[0])), key=lambda x: x[3])[:8]:
                print("    lowest: %-52s ref norm %.3g ratio %.3f cos %.4f n %d" % r)
        print("  %-13s n=%4d  norm ratio median %.3f [%.3f, %.3f]   cos median %.4f p10 %.4f min %.4f" % ((gname,) + stats[gname]))
    # Yardstick: the reference's OWN update_G on this fixture with every conv / norm / activation output and the gradient
    # flowing back through it rounded to bf16 (tests/devtools/measure_ref_jstep_quant.py jstep_640, dev container):
    #   encoder conv  cos median 0.9191 p10 0.9131   encoder bn  median 0.9198 p10 0.9025   (norm ratios 0.99)
    #   decoders      cos median 1.0000 p10 0.9611   painter     median 0.9975 p10 0.9939
    # i.e. ~0.08 of (1 - cos) in the encoder is what 16-bit storage costs ANY implementation of this step (the SIGM / L1
    # style terms back-propagate sign patterns).  Bound: (1 - cos) <= 1.4 x the yardstick's for the Masker (measured on
    # MI355X: 1.04-1.17 x), 2.5 x for the Painter (measured 2.2 x: the SPADE backward stores the re-materialised hidden map
    # and the gamma / beta gradient split in 16 bit, stores the emulation does not have).
    def within(stat, yard_median, yard_p10, slack):
        n, med_r, min_r, max_r, med_c, p10_c, min_c = stat
        assert 0.97 <= med_r <= 1.03, stat
        assert 1 - med_c <= slack * (1 - yard_median) + 1e-4 and 1 - p10_c <= slack * (1 - yard_p10) + 1e-4, stat

    # (jstep_small, same tool: encoder conv 0.9262 / 0.9206, bn 0.9255 / 0.9091, decoders 0.9998 / 0.9632, painter
    # 0.9954 / 0.9900)
    yard = {"jstep_640": {"encoder conv": (0.9191, 0.9131), "encoder bn": (0.9198, 0.9025), "decoders": (0.9995, 0.9611),
                          "painter": (0.9975, 0.9939)},
            "jstep_small": {"encoder conv": (0.9262, 0.9206), "encoder bn": (0.9255, 0.9091), "decoders": (0.9995, 0.9632),
                            "painter": (0.9954, 0.9900)}}[name]
    assert stats["encoder conv"][0] >= 100
    for grp in ("encoder conv", "encoder bn"):
        within(stats[grp], yard[grp][0], yard[grp][1], 1.4)
    # decoders: the tenth percentile is set by the depth decoder's three BatchNorm'd layers (enc4_1 / enc4_2 / enc4_3: the
    # same common-mode cancellation as the encoder's, cos 0.93-0.95 here against the encoder's 0.90).  Round 3 had loosened
    # this bound to 1.7 x (measured 1.45-1.53 x with fp32 atomics in the bias-gradient / loss-statistics reductions); round 4
    # made every reduction of the gradient path order-fixed (tests/test_gpu_determinism.py): (1 - p10) is now 1.13-1.27 x
    # the yardstick's, the same from run to run, and the bound is back at 1.4 x
    within(stats["decoders"], yard["decoders"][0], yard["decoders"][1], 1.4)
    if "p" in tasks:
        assert stats["painter"][0] >= 100
        within(stats["painter"], yard["painter"][0], yard["painter"][1], 2.5)
    sd = T.G.state_dict()
    for k in gold:
        if k.startswith("post.G."):
            ref, got = gold[k], sd[k[7:]].cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-2 * max(np.abs(ref).max(), 1e-3), k


def assert_d_side(T, gold, case, name, config, tasks):
    """After ``update_D``: the discriminator terms and the gradient norms / directions of every D tensor."""
    print("%s: D-side" % config)
    dmap = {"D.s.Advent": None, "D.m.Advent": None}
    for dom_task in ("s", "m"):
        ref = float(gold["D.%s.Advent" % dom_task][0])
        got = float(T.loss_log["D.%s.advent.r" % dom_task] + T.loss_log["D.%s.advent.s" % dom_task])
        print("  D.%s.Advent   reference %+.6g   hip %+.6g" % (dom_task, ref, got))
        assert abs(got - ref) <= 1e-2 * max(abs(ref), 1e-3)
    if "p" in tasks:
        ref, got = float(gold["D.p.gan"][0]), float(T.loss_log["D.p.gan"])
        print("  D.p.gan      reference %+.6g   hip %+.6g" % (ref, got))
        assert abs(got - ref) <= 1e-2 * abs(ref)
    rows = []
    _compare_grads(T.D, "D", gold, case["sub"], rows)
    # Yardstick (the same emulation run through update_D, jstep_640, dev container): D.p cos median 0.9907, D.s 0.9832,
    # D.m 0.9996.  All three see a generator that ExtraAdam has just moved by lr * g / (|g| + eps) ~ lr * sign(g) per
    # element, so sign flips of near-zero 16-bit generator gradients are part of their input noise.  Bound for D.p:
    # (1 - cos) <= 2.5 x the yardstick's (measured 1.6 x), for D.s 1.4 x.  (Round 3 had 1.6 x here: the value moved from
    # run to run with the fp32 atomics of the generator's bias gradients -- they decide the sign ExtraAdam gives near-zero
    # entries -- cos median 0.9760 - 0.9809 over seven runs of one build.  Round 4: no atomics on the gradient path, the
    # step is bit-reproducible (tests/test_gpu_determinism.py), measured 0.9814 / 0.9824 = 1.05 - 1.11 x, every run.)
    # D.m is the one discriminator whose gradient is a small difference of two large terms: the real and the simulated
    # call push the weights in opposite directions (labels 1 / 0 on near-identical entropy maps; |g_r + g_s| = 0.17 |g_r|),
    # and each call runs on its own w_bar / sigma (one power iteration per call), ROUNDED TO bf16 for the MFMA -- a
    # rounding the emulation above does not have.  tests/devtools/diag_advent_d.py jstep_640 (GPU box): the reference-exact
    # fp32 discriminator on the inputs this path produced reproduces the reference's gradients (cos 1.0000); rounding its
    # conv weights to bf16 per call costs cos 0.9558 (median; min 0.898), weights + activations 0.9677, and THAT variant
    # agrees with this path to 0.9999.  Bound: (1 - cos) <= 1.4 x (1 - 0.9677); measured 0.9606 - 0.9616.
    # (Round 2 found the real cause of the earlier 0.64 - 0.94 on D.s: the forward kept using the weights packed before
    # the optimizer step, so the D update saw the un-extrapolated generator; tests/test_gpu_train.py::
    # test_forward_uses_the_parameters_the_optimizer_wrote.)
    # On the 128 x 160 fixture the seg discriminator sees 32 x 40 entropy maps and its last bias gradient is a +-0.25 / N
    # cancellation between the two domains that rounds to exactly 0: emulation 0.94, measured 0.906 - 0.918.
    floors = ((("p.", 1 - 2.5 * (1 - 0.9907)), ("m.", 1 - 1.4 * (1 - 0.9677)), ("s.", 1 - 1.4 * (1 - 0.9832))) if name == "jstep_640" else
              (("p.", 0.985), ("m.", 0.98), ("s.", 0.89)))
    for grp, floor in floors:
        if not any(r[0].startswith(grp) for r in rows):
            continue
        st = _summ(rows, lambda k, grp=grp: k.startswith(grp))
        for r in rows:
            if r[0].startswith(grp) and (r[2] < 0.8 or r[3] < 0.8):
                print("    low: %-40s ref norm %.3g ratio %.3f cos %.3f" % (r[0], r[1], r[2], r[3]))
        print("  %-13s n=%4d  norm ratio median %.3f [%.3f, %.3f]   cos median %.4f p10 %.4f min %.4f" % (("D." + grp[0],) + st))
        assert 0.90 <= st[1] <= 1.05 and st[4] >= floor, (grp, st)


TRAJECTORY_TERMS = {  # golden key suffix -> Trainer.loss_log key(s) (summed)
    "G.task.d.s": ("G.d.s",), "G.task.s.crossent.s": ("G.s.crossent.s",), "G.task.m.bce.s": ("G.m.bce.s",),
    "G.task.m.minent.r": ("G.m.minent.r",), "G.task.m.advent.r": ("G.m.advent.r",),
    "G.p.vgg": ("G.p.vgg",), "G.p.featmatch": ("G.p.featmatch",), "G.p.gan": ("G.p.gan",),
    "D.p.gan": ("D.p.gan",), "D.s.Advent": ("D.s.advent.r", "D.s.advent.s"), "D.m.Advent": ("D.m.advent.r", "D.m.advent.s"),
}


def test_four_train_iterations_follow_the_reference_trajectory():
    """Four consecutive iterations of the training loop body on one batch (D frozen during the G update, ExtraAdam
    alternating extrapolation / step on both optimizers, step counter) vs the loss terms the REFERENCE's own loop logs
    (golden ``jstep_small``, keys it2.* .. it4.*: trainer.py:955-980 run four times).  Every iteration after the first sees
    parameters the optimizer wrote -- this is the test that would have caught the forward running on stale packed weights
    (the terms would simply not move).  The first ExtraAdam updates are lr * sign(g) per element, so 16-bit gradient noise
    on near-zero elements moves the trajectory a little (measured on MI355X: depth term 11.358 vs 11.319 after four iterations,
    cross-entropy 2.15704 vs 2.15687, VGG 402.35 vs 402.55, every other term to 3-4 digits); bounds: every term within 1.5 % of
    the reference's, and the CHANGE of the terms that move by more than 2 % over the four iterations (depth, cross-entropy,
    VGG, D.p) within 10 % of the reference's change (or 1 % of the term, for the ones that barely move: the ADVENT generator term changes by 2 %, and the fp32 atomics of the weight-gradient kernels move the fourth iteration's value by +-0.05 % from run to run)."""
    case = CASES_640["jstep_small"]
    gold = load_golden("jstep_small")
    T = _build_train(("d", "s", "m", "p"), case, 1)
    batch = _batch(case, 1, ("r", "s", "rf"))
    T.G.painter.set_latent_shape((case["B"], 3, case["H"], case["W"]), True)
    got = {}
    for it in range(1, case["iterations"] + 1):
        g, d = T.train_step(batch)
        assert torch.isfinite(g) and torch.isfinite(d)
        got[it] = {k: float(sum(T.loss_log[x] for x in v)) for k, v in TRAJECTORY_TERMS.items()}
    assert T.global_step == case["iterations"]
    print()
    for k in TRAJECTORY_TERMS:
        ref = [float(gold[("" if it == 1 else "it%d." % it) + k][0]) for it in range(1, case["iterations"] + 1)]
        mine = [got[it][k] for it in range(1, case["iterations"] + 1)]
        print("  %-22s reference %s\n  %-22s hip       %s" % (k, " ".join("%+.5f" % v for v in ref), "", " ".join("%+.5f" % v for v in mine)))
        for r, m in zip(ref, mine):
            assert abs(m - r) <= 1.5e-2 * max(abs(r), 1e-3), (k, ref, mine)
        if abs(ref[-1] - ref[0]) > 2e-2 * abs(ref[0]):
            assert abs((mine[-1] - mine[0]) - (ref[-1] - ref[0])) <= max(0.10 * abs(ref[-1] - ref[0]), 1e-2 * abs(ref[0])), (k, ref, mine)


def test_sixty_iterations_stay_finite_and_learn():
    """A longer run of the joint training loop on one batch (small fixture): every logged term stays finite, the supervised
    terms (depth, segmentation cross-entropy, mask BCE, VGG) fall, parameters and running statistics of every network move,
    and the packed-weight caches follow (the forward of the trained generator equals that of a fresh one loaded from its
    state dict -- the property the stale-weights bug broke)."""
    from climategan_amd.generator import create_generator

    case = CASES_640["jstep_small"]
    T = _build_train(("d", "s", "m", "p"), case, 1)
    batch = _batch(case, 1, ("r", "s", "rf"))
    T.G.painter.set_latent_shape((case["B"], 3, case["H"], case["W"]), True)
    first = None
    for it in range(60):
        g, d = T.train_step(batch)
        assert torch.isfinite(g) and torch.isfinite(d), it
        if first is None:
            first = {k: float(v) for k, v in T.loss_log.items()}
    last = {k: float(v) for k, v in T.loss_log.items()}
    assert all(v == v and abs(v) < 1e6 for v in last.values()), last
    print("\n  after 60 iterations: " + ", ".join("%s %.4g -> %.4g" % (k, first[k], last[k])
                                                 for k in ("G.d.s", "G.s.crossent.s", "G.m.bce.s", "G.p.vgg", "D.p.gan")))
    for k, factor in (("G.d.s", 0.5), ("G.s.crossent.s", 0.9), ("G.m.bce.s", 0.99), ("G.p.vgg", 0.95)):
        assert last[k] < factor * first[k], (k, first[k], last[k])
    sd = {k: v.detach().clone() for k, v in T.G.state_dict().items()}
    fresh = create_generator(T.opts, device="cuda", no_init=True)
    fresh.load_state_dict(sd)
    fresh.set_compute_dtype(torch.bfloat16)
    fresh.decoders["d"]._target_size = case["W"] // 4
    fresh.decoders["s"].set_target_size((case["H"] // 4, case["W"] // 4))
    T.G.eval(); fresh.eval()
    with torch.no_grad():
        a, b = T.G.masker_forward(batch["r"]["data"]["x"]), fresh.masker_forward(batch["r"]["data"]["x"])
    for k in a:
        assert torch.equal(a[k], b[k]), k
