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
        if name == "painter_640" or case["kind"] in ("extra_adam", "masker", "infer", "dstep_p", "gstep_p", "cloudy", "maskspade", "masker_losses", "mstep"):
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
