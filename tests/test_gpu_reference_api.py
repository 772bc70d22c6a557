"""The drop-in boundary under TRAINING (north_star: "trainer.py and apply_events.py call into it unchanged").

The reference's trainer drives the model API with NCHW tensors under autograd: ``G.paint(m, x)``, ``G.decoders[t](z)``,
``D["p"](cat)`` return tensors that it feeds to torch expressions and to the loss classes, and it casts the modules with
``G.half()`` (apply_events.py:467-468).  ``tests/ref_call_pattern.py`` restates exactly that call pattern; here it runs one
G update and one D update on the small joint fixture and must reproduce the REFERENCE's own step (golden ``jstep_small``:
``Trainer.update_G`` + ``Trainer.update_D``) within the bounds the package's own NHWC trainer is held to -- loss terms,
per-tensor gradient norms and directions, BatchNorm running statistics.  Also: the layout Functions' gradients, the module
casts, and the two non-default Masker options (pl4m, ``gen.m.use_dada``) against their own golden (``jstep_pl4m_small``),
through both the reference call pattern and the package's trainer."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle.make_golden_640 import CASES_640
from ref_call_pattern import ReferenceCalls
from test_gpu_configs_640 import MASKER_LOSS_KEYS, _batch, _build_train, _check_terms, _compare_grads, _summ, assert_d_side, assert_g_side

pytestmark = pytest.mark.gpu


def _small(name="jstep_small", **opt_overrides):
    case = CASES_640[name]
    T = _build_train(("d", "s", "m", "p"), case, 1, opt_overrides=opt_overrides)
    T.G.painter.set_latent_shape((case["B"], 3, case["H"], case["W"]), True)
    return case, T, _batch(case, 1, ("r", "s", "rf"))


def test_reference_call_pattern_reproduces_the_reference_step():
    """update_G / update_D written with the reference's signatures only (NCHW tensors, torch glue, ``G.bfloat16()``)."""
    case, T, batch = _small()
    gold = load_golden("jstep_small")
    # the reference's cast (apply_events.py:467-468 uses .half(); training needs bf16's range)
    assert T.G.bfloat16() is T.G and T.D.bfloat16() is T.D
    assert all(p.dtype == torch.float32 for p in T.G.parameters()), "the fp32 masters must survive the cast"
    R = ReferenceCalls(T)
    g_loss = R.update_G(batch, step=0)
    assert torch.isfinite(g_loss)
    assert abs(float(g_loss) - float(gold["G.total_loss"][0])) <= 2e-2 * abs(float(gold["G.total_loss"][0]))
    assert_g_side(T, gold, case, "jstep_small", "reference_call_pattern", ("d", "s", "m", "p"))
    d_loss = R.update_D(batch, step=0)
    assert torch.isfinite(d_loss)
    assert abs(float(d_loss) - float(gold["D.total_loss"][0])) <= 1e-2 * abs(float(gold["D.total_loss"][0]))
    assert_d_side(T, gold, case, "jstep_small", "reference_call_pattern", ("d", "s", "m", "p"))


PL4M_TERMS = dict(MASKER_LOSS_KEYS, **{"G.task.m.pl4m.r": "G.m.pl4m.r"})


@pytest.mark.parametrize("driver", ["package_trainer", "reference_call_pattern"])
def test_pl4m_and_mask_dada_match_the_reference_step(driver):
    """``painter_loss_for_masker`` (trainer.py:1618-1651) and ``gen.m.use_dada`` (trainer.py:1566-1570, blocks.py:304-305)
    vs the reference's own ``update_G`` / ``update_D`` with both switched on (golden ``jstep_pl4m_small``).  pl4m sends a
    gradient from the Painter's discriminator back into the Masker through the paste, the discriminator's mask channel
    and the Painter's conditioning image; it also spends one extra spectral-norm power iteration on the Painter and on D
    before the Painter's own loss, which the VGG / GAN / feature-matching terms see (419.1 -> 429.2 in the reference)."""
    case, T, batch = _small("jstep_pl4m_small", m_use_dada=True)
    gold = load_golden("jstep_pl4m_small")
    T.use_pl4m = True
    if driver == "package_trainer":
        T.update_G(batch)
    else:
        R = ReferenceCalls(T)
        R.use_pl4m = True
        R.update_G(batch, step=0)
    print("\n%s: G-side loss terms (pl4m + gen.m.use_dada)" % driver)
    _check_terms(T, gold, PL4M_TERMS, 3e-2, driver)
    _check_terms(T, gold, {"G.p.vgg": "G.p.vgg", "G.p.gan": "G.p.gan", "G.p.featmatch": "G.p.featmatch"}, 2e-2, driver)
    rows = []
    _compare_grads(T.G, "G", gold, case["sub"], rows)
    is_conv = lambda k: k.endswith("weight_bar") or (k.endswith(".weight") and ".bn" not in k and ".norm" not in k)
    for gname, sel, floor in (("encoder conv", lambda k: k.startswith("encoder.") and is_conv(k), 0.88),
                              ("decoders.m", lambda k: k.startswith("decoders.m."), 0.97),
                              ("decoders", lambda k: k.startswith("decoders."), 0.99),
                              ("painter", lambda k: k.startswith("painter."), 0.98)):
        st = _summ(rows, sel)
        print("  %-13s n=%4d  norm ratio median %.3f [%.3f, %.3f]   cos median %.4f p10 %.4f min %.4f" % ((gname,) + st))
        assert st[0] > 0 and 0.95 <= st[1] <= 1.05 and st[4] >= floor, (gname, st)
    if driver == "package_trainer":
        T.update_D(batch)
    else:
        R.update_D(batch, step=0)
    for task in ("s", "m"):
        ref = float(gold["D.%s.Advent" % task][0])
        got = float(T.loss_log["D.%s.advent.r" % task] + T.loss_log["D.%s.advent.s" % task])
        print("  D.%s.Advent   reference %+.6g   hip %+.6g" % (task, ref, got))
        assert abs(got - ref) <= 1e-2 * abs(ref)


def test_layout_functions_carry_gradients():
    """``ToNchwFn`` / ``FromNchwFn`` / ``FromNchwPairFn`` against the same expressions in torch."""
    from climategan_amd import functional as Fn
    from climategan_amd import ops

    g = torch.Generator(device="cuda").manual_seed(5)
    n, c, h, w = 2, 3, 12, 20
    for dt in (torch.float16, torch.bfloat16):
        y = torch.zeros((n, h, w, 8), dtype=dt, device="cuda")
        y[..., :c] = torch.randn((n, h, w, c), device="cuda", generator=g).to(dt)
        y.requires_grad_(True)
        x = torch.randn((n, c, h, w), device="cuda", generator=g)
        m = (torch.rand((n, 1, h, w), device="cuda", generator=g) > 0.4).float()
        wgt = torch.randn((n, c, h, w), device="cuda", generator=g)
        out = Fn.to_nchw(ops.NHWC(y, c), paste_x=x, paste_m=m)
        ref = x * (1 - m) + y[..., :c].float().permute(0, 3, 1, 2) * m
        assert torch.equal(out, ref)
        (out * wgt).sum().backward()
        want = torch.zeros_like(y)
        want[..., :c] = (wgt * m).permute(0, 2, 3, 1).to(dt)
        assert torch.equal(y.grad, want)

        xin = torch.randn((n, c, h, w), device="cuda", generator=g, requires_grad=True)
        z = Fn.from_nchw(xin, dt, mask=m)
        assert z.t.requires_grad and z.c == c and z.t.shape == (n, h, w, 8)
        assert torch.equal(z.t[..., :c].float(), (xin.detach() * (1 - m)).permute(0, 2, 3, 1).to(dt).float())
        gz = torch.randn(z.t.shape, device="cuda", generator=g).to(dt)
        z.t.backward(gz)
        assert torch.equal(xin.grad, gz[..., :c].float().permute(0, 3, 1, 2) * (1 - m))

        big = (100 + 50 * torch.rand((n, c, h, w), device="cuda", generator=g)).requires_grad_(True)
        pair = Fn.from_nchw_pair(big, dt)
        assert pair.c == 2 * c
        val = pair.t[..., :c].float() + pair.t[..., c:2 * c].float()
        assert (val - big.detach().permute(0, 2, 3, 1)).abs().max() <= (2e-4 if dt == torch.float16 else 2e-3)
        gp = torch.randn(pair.t.shape, device="cuda", generator=g).to(dt)
        pair.t.backward(gp)
        assert torch.equal(big.grad, gp[..., :c].float().permute(0, 3, 1, 2))


def test_module_casts_select_the_compute_type_and_keep_fp32_masters():
    """``G.half()`` / ``.bfloat16()`` / ``.to(dtype)`` / ``.float()`` (apply_events.py:467-468) and the same
    on ``OmniDiscriminator``; ``G.half()`` + ``infer_all(half=True)`` is bit-identical to ``set_compute_dtype(fp16)``."""
    from climategan_amd import fill
    from climategan_amd.bn_fusion import bn_fuse
    from climategan_amd.config import default_opts
    from climategan_amd.discriminator import create_discriminator
    from climategan_amd.trainer import Trainer

    opts = default_opts()
    opts.tasks = ["d", "s", "m", "p"]
    opts.gen.p.latent_dim, opts.gen.p.spade_n_up = 32, 4
    T = Trainer(opts, device="cuda").setup(inference=True)
    G = T.G
    sd = {k: v.clone() for k, v in G.state_dict().items()}
    x = torch.from_numpy(fill.uniform((2, 3, 128, 160), 11)).cuda()
    T.G.decoders["d"]._target_size = 40
    T.G.decoders["s"].set_target_size((32, 40))

    def run():
        G.load_state_dict(sd)                              # same spectral-norm u / v for every run
        return T.infer_all(x, numpy=True, half=True, bin_value=0.5, ignore_event={"wildfire"}, return_masks=True)

    G.set_compute_dtype(torch.float16)
    base = run()
    assert bn_fuse(G) is G                                 # apply_events.py:465-466: folding happens at pack time here
    assert G.bfloat16() is G and G.compute_dtype == torch.bfloat16 and G.painter.compute_dtype == torch.bfloat16
    other = run()
    assert any(not np.array_equal(base[k], other[k]) for k in base)          # the cast did switch the kernels' type
    assert G.half() is G and G.compute_dtype == torch.float16 and G.encoder.compute_dtype == torch.float16
    assert all(p.dtype == torch.float32 for p in G.parameters()) and all(b.dtype != torch.float16 for b in G.buffers())
    again = run()
    for k in base:
        assert np.array_equal(base[k], again[k]), k
    assert G.to(torch.bfloat16) is G and G.compute_dtype == torch.bfloat16
    assert G.float() is G and G.compute_dtype == torch.bfloat16 and next(G.parameters()).dtype == torch.float32
    assert G.to("cuda").to(device="cuda", dtype=torch.float16).compute_dtype == torch.float16
    D = create_discriminator(opts, "cuda", no_init=True)
    assert D.half() is D and D["p"].compute_dtype == torch.float16 and D["m"]["Advent"].compute_dtype == torch.float16
    assert D.bfloat16()["p"].discriminator_0.compute_dtype == torch.bfloat16
    assert all(p.dtype == torch.float32 for p in D.parameters())
