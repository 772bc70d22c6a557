"""Direct comparison oracle.cpu_ref <-> the real reference, imported from /root/reference.
Dev-container only (skipped on the GPU box, where the reference does not exist)."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, ref_shim

pytestmark = pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")

from helpers import case_state_dict, golden_cases, run_oracle, t  # noqa: E402


def test_golden_files_are_current():
    """The committed fixtures equal what the reference produces now (generator script is in sync)."""
    from helpers import load_golden
    from oracle.make_golden import run_reference

    cases = golden_cases()
    for name in ("spade_c20", "resblk_16_8", "painter_up4", "disc_fc"):
        ref = run_reference(name, cases[name])
        gold = load_golden(name)
        for k in gold:
            np.testing.assert_allclose(ref[k], gold[k], rtol=0, atol=1e-6)


def test_second_forward_uses_updated_uv():
    """Spectral norm mutates u/v on every forward (norms.py:100-112,141-143): two consecutive forwards differ
    and the oracle tracks the reference through both."""
    from oracle.make_golden import build_reference_module, case_inputs

    cases = golden_cases()
    name = "resblk_16_8"
    mod, _ = build_reference_module(cases[name])
    inp = {k: t(v) for k, v in case_inputs(name, cases[name]).items()}
    sd = {"b." + k: v for k, v in case_state_dict(cases[name]).items()}
    with torch.no_grad():
        r1 = mod(inp["x"], inp["seg"])
        r2 = mod(inp["x"], inp["seg"])
        o1 = cpu_ref.spade_resnet_block(inp["x"], inp["seg"], sd, "b")
        o2 = cpu_ref.spade_resnet_block(inp["x"], inp["seg"], sd, "b")
    assert (r1 - r2).abs().max() > 1e-6
    assert (r1 - o1).abs().max() < 2e-5
    assert (r2 - o2).abs().max() < 2e-5


def test_default_discriminator_shapes():
    """Default D_p (ndf 64, n_layers 4, num_D 3): oracle output list structure equals the reference's."""
    disc = ref_shim.ref("discriminator")
    mod = disc.define_D(input_nc=4, ndf=8, n_layers=4, norm="instance", use_sigmoid=False,
                        get_intermediate_features=True, num_D=3)
    from climategan_amd import fill
    from helpers import disc_p_shapes

    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    assert shapes == disc_p_shapes(4, 8, 4, 3)
    sd_np = fill.fill_state_dict(shapes, 5)
    mod.load_state_dict({k: t(v) for k, v in sd_np.items()})
    x = t(fill.uniform((1, 4, 192, 224), 77))
    with torch.no_grad():
        ref = mod(x)
        got = cpu_ref.multiscale_discriminator(x, {k: t(v) for k, v in sd_np.items()}, 3, 4)
    assert len(ref) == len(got) == 3
    for a, b in zip(ref, got):
        assert len(a) == len(b) == 6
        for fa, fb in zip(a, b):
            assert fa.shape == fb.shape
            assert (fa - fb).abs().max() < 2e-5


def test_state_dict_shapes_match_reference():
    from helpers import module_shapes
    from oracle.make_golden import build_reference_module

    for name, case in golden_cases().items():
        if name == "painter_640" or case["kind"] not in ("spade", "resblk", "painter", "paint", "disc_p", "disc_fc"):
            continue
        mod, _ = build_reference_module(case)
        assert {k: tuple(v.shape) for k, v in mod.state_dict().items()} == module_shapes(case), name


def test_infer_state_dict_layout_matches_reference_trainer():
    """The restated full-generator layout (masker + painter) used by the infer fixture == the reference Trainer's G."""
    from helpers import infer_shapes
    from oracle.make_golden import reference_trainer

    case = golden_cases()["infer_small"]
    _, shapes = reference_trainer(case)
    assert shapes == infer_shapes(case)


def test_reference_resume_reads_a_checkpoint_saved_by_the_mirror(tmp_path):
    """The other direction of tests/test_checkpoint.py: ``climategan_amd.Trainer.save`` writes, the reference's own
    ``Trainer.resume`` (trainer.py:422-579) reads -- strict G / D loads, optimizer state accepted, counters restored."""
    import contextlib
    import io
    from types import SimpleNamespace

    from oracle.make_golden_ckpt import build, small_opts
    from test_checkpoint import FIX, make_trainer

    T = make_trainer(tmp_path)
    with pytest.warns(UserWarning):
        T._resolve_checkpoint = lambda: torch.load(FIX / "checkpoints" / "latest_ckpt.pth", weights_only=False)
        T.resume()
    T.epoch, T.global_step = 4, 22
    T.save()
    opts = small_opts(ref_shim.default_opts())
    opts.output_path = str(tmp_path)
    tr = ref_shim.ref("trainer")
    G, D, g_opt, g_sched, d_opt, d_sched = build(ref_shim, opts)
    ns = SimpleNamespace(opts=opts, device=torch.device("cpu"), logger=SimpleNamespace(epoch=0, global_step=0), G=G, D=D,
                         g_opt=g_opt, d_opt=d_opt, g_scheduler=g_sched, d_scheduler=d_sched,
                         exp=SimpleNamespace(log_text=lambda *a, **k: None))
    ns.update_learning_rates = lambda: tr.Trainer.update_learning_rates(ns)
    with contextlib.redirect_stdout(io.StringIO()):
        tr.Trainer.resume(ns)
    assert ns.logger.epoch == 4 and ns.logger.global_step == 22
    for mine, theirs in ((T.G, G), (T.D, D)):
        a, b = mine.state_dict(), theirs.state_dict()
        assert list(a) == list(b)
        assert all(torch.equal(a[k], b[k]) for k in a)
    params = [p for g in g_opt.param_groups for p in g["params"]]
    mine = [p for g in T.g_opt.param_groups for p in g["params"]]
    for i, p in enumerate(params):
        if p.requires_grad:
            assert torch.equal(g_opt.state[p]["exp_avg"], T.g_opt.state[mine[i]]["exp_avg"])
            assert g_opt.state[p]["step"] == T.g_opt.state[mine[i]]["step"]


def test_get_optimizer_groups_match_reference():
    """Per-task learning rates (optim.py:82-108): same groups, same parameter order, same rates, same lr_names."""
    import contextlib
    import io

    from climategan_amd.config import default_opts
    from climategan_amd.generator import create_generator
    from climategan_amd.optim import get_optimizer

    ropts = ref_shim.default_opts()
    ropts.tasks = ["m", "s", "d"]
    ropts.gen.opt.lr = {"default": 5e-5, "m": 1e-4, "s": 2e-4}
    with contextlib.redirect_stdout(io.StringIO()):
        RG = ref_shim.ref("generator").create_generator(ropts, "cpu", no_init=True)
    r_opt, r_sched, r_names = ref_shim.ref("optim").get_optimizer(RG, ropts.gen.opt, ropts.tasks)
    o = default_opts()
    o.tasks = ["m", "s", "d"]
    o.gen.opt.lr = {"default": 5e-5, "m": 1e-4, "s": 2e-4}
    G = create_generator(o, device="cpu", no_init=True)
    opt, sched, names = get_optimizer(G, o.gen.opt, o.tasks)
    assert names == r_names == ["encoder", "decoder_m", "decoder_s", "decoder_d"]
    assert [g["lr"] for g in opt.param_groups] == [g["lr"] for g in r_opt.param_groups] == [1e-4, 1e-4, 2e-4, 5e-5]
    assert [[tuple(p.shape) for p in g["params"]] for g in opt.param_groups] == \
           [[tuple(p.shape) for p in g["params"]] for g in r_opt.param_groups]
    assert type(sched).__name__ == type(r_sched).__name__ == "StepLR"
    assert sched.step_size == r_sched.step_size and sched.gamma == r_sched.gamma
