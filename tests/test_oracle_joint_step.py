"""Pin ``oracle.cpu_ref.joint_train_step`` (the CPU restatement of Trainer.update_G + Trainer.update_D that
``bench.py`` times as ``cpu_baseline``) against the golden produced by the REFERENCE's own ``update_G`` / ``update_D``
(tests/golden/jstep_small.npz, oracle/make_golden_640.py): every logged loss term, the gradient norm and a seeded
sub-sample of the gradient of every trainable G and D tensor.  CPU-only, runs everywhere."""
import numpy as np
import torch

from climategan_amd import fill
from helpers import load_golden, masker_shapes, painter_shapes, disc_p_shapes, disc_fc_shapes, t
from oracle import cpu_ref
from oracle.make_golden import grad_subsample
from oracle.make_golden_640 import CASES_640, generator_fill, jstep_inputs


def _shapes(case):
    g = dict(masker_shapes())
    g.update({"painter." + k: v for k, v in painter_shapes(case["latent_dim"], case["n_up"]).items()})
    d = {"p." + k: v for k, v in disc_p_shapes(4, case["ndf"], case["n_layers"], 3).items()}
    d.update({"m.Advent." + k: v for k, v in disc_fc_shapes(2).items()})
    d.update({"s.Advent." + k: v for k, v in disc_fc_shapes(11).items()})
    return g, d


def test_joint_train_step_matches_reference_update_g_and_update_d():
    case = CASES_640["jstep_small"]
    gold = load_golden("jstep_small")
    gs, ds = _shapes(case)
    sd_g = {k: t(v) for k, v in generator_fill(gs, case).items()}
    sd_d = {k: t(v) for k, v in fill.fill_state_dict(ds, case["seed"] + 1).items()}
    sd_v = {k: t(v) for k, v in fill.fill_state_dict(cpu_ref.vgg19_shapes(), case["vgg_seed"], gain=case["vgg_gain"]).items()}
    batch = {dom: {k: t(v) for k, v in d.items()} for dom, d in jstep_inputs(case).items()}
    torch.set_num_threads(8)
    out = cpu_ref.joint_train_step(sd_g, sd_d, sd_v, batch, case["n_up"], 3, case["n_layers"])
    terms = out["terms"]
    for k in gold:
        if k.startswith(("G.", "D.")) and k in terms:
            ref, got = float(gold[k][0]), float(terms[k])
            assert abs(got - ref) <= 2e-4 * max(abs(ref), 1e-2), (k, got, ref)
    checked = 0
    for side, grads in (("G", out["g_grads"]), ("D", out["d_grads"])):
        keys = [k[len("gnorm.%s." % side):] for k in gold if k.startswith("gnorm.%s." % side)]
        assert set(keys) == set(grads), (side, set(keys) ^ set(grads))
        for key in keys:
            ref_n = float(gold["gnorm.%s.%s" % (side, key)][0])
            g = grads[key]
            a = gold["gsub.%s.%s" % (side, key)].astype(np.float64)
            b = grad_subsample(key, g, case["sub"]).astype(np.float64)
            base = key.rsplit(".", 1)[0]
            sib = max([float(gold[n][0]) for n in ("gnorm.%s.%s.weight" % (side, base), "gnorm.%s.%s.weight_bar" % (side, base))
                       if n in gold] + [0.0])
            if ref_n < 1e-4 * max(sib, 1e-2):     # a bias in front of a norm layer: zero true gradient, fp32 noise
                assert float(g.norm()) < 1e-3 * max(sib, 1e-2), key
                continue
            # G side: two fp32 code paths of the same arithmetic.  D side: the D update runs on the EXTRAPOLATED generator,
            # whose ExtraAdam step is lr * g / (|g| + eps) ~ lr * sign(g) per element -- fp32 noise on near-zero G
            # gradient elements flips signs, i.e. moves those parameters by 2 lr, and the D gradients inherit that
            ntol, ctol = (2e-3, 0.9995) if side == "G" else (1e-2, 0.995)
            assert abs(float(g.norm()) / ref_n - 1) < ntol, (side, key, float(g.norm()), ref_n)
            cos = (a * b).sum() / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300)
            assert cos > ctol, (side, key, cos)
            checked += 1
    assert checked > 500
    for k in gold:
        if k.startswith("post.G."):
            np.testing.assert_allclose(out["bn_after_update_G"][k[7:]].numpy(), gold[k], rtol=1e-4, atol=1e-6)
