"""Seeded random-configuration sweep at MODULE level against the oracle (oracle.cpu_ref, pinned on the reference by the
golden fixtures): the Painter (``OmniGenerator.paint``) and the multi-scale PatchGAN discriminator at random widths,
depths, batch sizes and ragged image extents -- the shape-dependent host logic (latent shape, folded upsamples, tile
grids, pooled scales) that fixed golden shapes cannot reach.  Bounds: the 16-bit allowances of tests/test_gpu_painter.py
/ test_gpu_discriminator.py (twice the reference's own half-precision deviation)."""
import numpy as np
import pytest
import torch

from climategan_amd import fill
from helpers import disc_p_shapes, painter_shapes, t
from oracle import cpu_ref

pytestmark = pytest.mark.gpu


def painter_cases(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        n_up = int(rng.choice([2, 3, 4]))
        zh, zw = int(rng.choice([1, 2, 3, 5])), int(rng.choice([1, 2, 4, 7]))
        out.append(dict(latent_dim=int(rng.choice([16, 24, 32])), n_up=n_up, B=int(rng.choice([1, 2])),
                        H=zh * 2 ** n_up, W=zw * 2 ** n_up, seed=int(rng.randint(1, 10000))))
    return out


@pytest.mark.parametrize("case", painter_cases(8, 5))
def test_paint_random_configurations(case):
    from climategan_amd.config import default_opts
    from climategan_amd.generator import create_generator

    opts = default_opts()
    opts.tasks = ["p"]
    opts.gen.p.latent_dim, opts.gen.p.spade_n_up = case["latent_dim"], case["n_up"]
    G = create_generator(opts, device="cuda")
    sd = {k: t(v) for k, v in fill.fill_state_dict(painter_shapes(case["latent_dim"], case["n_up"]), seed=case["seed"]).items()}
    G.painter.load_state_dict(sd, strict=True)
    G.set_compute_dtype(torch.float16)
    B, H, W = case["B"], case["H"], case["W"]
    G.painter.set_latent_shape((B, 3, H, W), True)
    x = t(fill.uniform((B, 3, H, W), case["seed"] + 1))
    m = t(fill.rect_mask(B, H, W, case["seed"] + 2))
    zh, zw = H // 2 ** case["n_up"], W // 2 ** case["n_up"]
    assert (G.painter.z_h, G.painter.z_w) == (zh, zw)
    with torch.no_grad():
        ref = cpu_ref.paint(sd, m, x, zh, zw)
        got = G.paint(m.cuda(), x.cuda()).cpu()
    err = (got - ref).abs()
    assert err.max().item() <= 2.5e-2 and err.mean().item() <= 2.5e-3, (err.max().item(), err.mean().item())
    outside = (m == 0).expand_as(x)
    assert torch.equal(got[outside], x[outside])
    for k, v in G.painter.state_dict().items():          # one power iteration on both sides
        if k.endswith("weight_u"):
            assert (v.cpu() - sd[k]).abs().max() < 2e-5, k


def disc_cases(n, seed):
    rng = np.random.RandomState(seed)
    return [dict(ndf=int(rng.choice([4, 8, 16])), n_layers=int(rng.choice([2, 3, 4])), num_D=int(rng.choice([1, 2, 3])),
                 B=int(rng.choice([1, 2])), H=int(rng.choice([64, 96, 130, 161])), W=int(rng.choice([64, 100, 128, 175])),
                 seed=int(rng.randint(1, 10000))) for _ in range(n)]


@pytest.mark.parametrize("case", disc_cases(8, 6))
def test_multiscale_discriminator_random_configurations(case):
    from climategan_amd.discriminator import define_D

    D = define_D(input_nc=4, ndf=case["ndf"], n_layers=case["n_layers"], norm="instance", use_sigmoid=False,
                 get_intermediate_features=True, num_D=case["num_D"]).cuda()
    shapes = disc_p_shapes(4, case["ndf"], case["n_layers"], case["num_D"])
    assert {k: tuple(v.shape) for k, v in D.state_dict().items()} == shapes
    sd = {k: t(v) for k, v in fill.fill_state_dict(shapes, seed=case["seed"]).items()}
    D.load_state_dict(sd, strict=True)
    D.compute_dtype = torch.float16
    x = t(fill.uniform((case["B"], 4, case["H"], case["W"]), case["seed"] + 3))
    with torch.no_grad():
        try:
            ref = cpu_ref.multiscale_discriminator(x, sd, case["num_D"], case["n_layers"])
        except RuntimeError:
            # the coarsest scale is smaller than a 4x4 kernel: torch refuses it, and so must the HIP path
            with pytest.raises(RuntimeError):
                D(x.cuda())
            return
        got = D(x.cuda())
    assert len(got) == len(ref) == case["num_D"]
    for a, b in zip(got, ref):
        assert len(a) == len(b) == case["n_layers"] + 2
        for fa, fb in zip(a, b):
            assert fa.shape == fb.shape
            scale = max(fb.abs().max().item(), 1e-6)
            assert (fa.cpu() - fb).abs().max().item() <= 2e-2 * scale, (tuple(fb.shape), scale)


@pytest.mark.parametrize("H,W,B,seed", [(72, 104, 1, 301), (96, 64, 2, 302), (120, 88, 1, 303)])
def test_masker_other_extents(H, W, B, seed):
    """The full default Masker (ResNet-101 OS 8, DADA depth, DeepLab-v3+ seg with its 82x82-style padded ASPP output,
    mask decoder) at extents other than the golden fixture's 128x160, against the oracle: feature-map sizes of strided
    convs on non-multiples of 16, align_corners resizes to other targets, the mask decoder's x8 upsampling."""
    from climategan_amd.config import default_opts
    from climategan_amd.generator import create_generator
    from helpers import masker_state_dict

    case = {"seed": seed, "gain": 1.6}
    opts = default_opts()
    opts.tasks = ["d", "s", "m"]
    G = create_generator(opts, device="cuda")
    sd = masker_state_dict(case)
    G.load_state_dict(sd, strict=True)
    G.eval()
    G.set_compute_dtype(torch.float16)
    G.decoders["d"]._target_size = W // 4
    G.decoders["s"].set_target_size((H // 4, W // 4))
    x = t(fill.uniform((B, 3, H, W), seed + 5))
    with torch.no_grad():
        ref = cpu_ref.masker_forward(sd, x, (H // 4, W // 4), d_target=W // 4)
        got = G.masker_forward(x.cuda())
    for k in ("d", "s", "m"):
        a, b = got[k].cpu(), ref[k]
        assert a.shape == b.shape, (k, a.shape, b.shape)
        scale = max(b.abs().max().item(), 1e-6)
        err = (a - b).abs()
        assert err.max().item() <= 3e-2 * scale and err.mean().item() <= 4e-3 * scale, (k, err.max().item(), scale)
