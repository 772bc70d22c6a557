"""GPU parity of the discriminator modules (HIP, through the module API) against the golden vectors produced by the
real reference.  Bound: 16-bit path vs fp32 reference, relative to each feature map's scale (2e-2 fp16 max /
3e-3 mean; the PatchGAN has 4-6 spectral-norm convs + instance norms in sequence)."""
import numpy as np
import pytest
import torch

from helpers import case_state_dict, golden_cases, load_golden, t
from oracle.make_golden import case_inputs

pytestmark = pytest.mark.gpu
CASES = golden_cases()


def _check(got, ref, name):
    scale = max(np.abs(ref).max(), 1e-6)
    err = np.abs(got - ref)
    assert err.max() <= 2e-2 * scale, "%s: max err %.3g (scale %.3g)" % (name, err.max(), scale)
    assert err.mean() <= 3e-3 * scale, "%s: mean err %.3g (scale %.3g)" % (name, err.mean(), scale)


def test_multiscale_patchgan_matches_reference_golden():
    from climategan_amd.discriminator import define_D

    case = CASES["disc_p"]
    gold = load_golden("disc_p")
    D = define_D(input_nc=4, ndf=case["ndf"], n_layers=case["n_layers"], norm="instance", use_sigmoid=False,
                 get_intermediate_features=True, num_D=case["num_D"]).cuda()
    D.load_state_dict(case_state_dict(case), strict=True)
    x = t(case_inputs("disc_p", case)["x"]).cuda()
    with torch.no_grad():
        res = D(x)
    assert len(res) == case["num_D"]
    for i, scale_out in enumerate(res):
        assert len(scale_out) == case["n_layers"] + 2
        for j, f in enumerate(scale_out):
            ref = gold["d%d_%d" % (i, j)]
            assert tuple(f.shape) == ref.shape
            _check(f.cpu().numpy(), ref, "d%d_%d" % (i, j))
    sd = D.state_dict()
    for k in gold:
        if k.startswith("post."):
            assert np.abs(sd[k[5:]].cpu().numpy() - gold[k]).max() < 2e-5, k


def test_advent_fc_discriminator_matches_reference_golden():
    from climategan_amd.discriminator import get_fc_discriminator

    case = CASES["disc_fc"]
    gold = load_golden("disc_fc")
    D = get_fc_discriminator(num_classes=case["num_classes"], use_norm=True).cuda()
    D.load_state_dict(case_state_dict(case), strict=True)
    x = t(case_inputs("disc_fc", case)["x"]).cuda()
    with torch.no_grad():
        y = D(x)
    assert tuple(y.shape) == gold["y"].shape
    _check(y.cpu().numpy(), gold["y"], "fc")
    sd = D.state_dict()
    for k in gold:
        if k.startswith("post."):
            assert np.abs(sd[k[5:]].cpu().numpy() - gold[k]).max() < 2e-5, k


def test_omni_discriminator_layout_and_default_shapes():
    """Default D (ndf 64, n_layers 4, num_D 3) on a 2x4x256x256 input: 3 scales x 6 maps, reference shapes."""
    from climategan_amd.config import default_opts
    from climategan_amd.discriminator import create_discriminator

    D = create_discriminator(default_opts(), "cuda")
    assert len(D.state_dict()) == 112   # SURVEY 8b [probe]
    x = torch.rand(2, 4, 256, 256, device="cuda") * 2 - 1
    with torch.no_grad():
        res = D["p"](x)
        ym = D["m"]["Advent"](torch.rand(2, 2, 256, 256, device="cuda"))
        ys = D["s"]["Advent"](torch.rand(2, 11, 64, 64, device="cuda"))
    assert [tuple(f.shape[-2:]) for f in res[0]] == [(128, 128), (64, 64), (32, 32), (16, 16), (15, 15), (14, 14)]
    assert [tuple(f.shape[-2:]) for f in res[2]] == [(32, 32), (16, 16), (8, 8), (4, 4), (3, 3), (2, 2)]
    assert tuple(ym.shape) == (2, 1, 8, 8) and tuple(ys.shape) == (2, 1, 2, 2)
    assert all(torch.isfinite(f).all() for s in res for f in s)
