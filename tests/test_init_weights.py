"""A22: ``init_weights`` and the create_* initialisation rules (reference tutils.py:26-85, generator.py:24-61,
discriminator.py:16-39, deeplab/deeplab_v3.py:59-64,111-116,178-190, deeplab/__init__.py:43-67).

Two layers: (1) everywhere -- the six ``init_type`` draws hit the formulas' standard deviations and only the modules the
class-name rule selects; (2) dev container -- the mirror and the REAL reference, built from the same options, call the
same ``torch.nn.init`` functions on the same state-dict entries in the same order (recorded by wrapping ``torch.nn.init``),
including the pretrained DeepLab checkpoint both load when ``no_init`` is False."""
import math

import pytest
import torch
import torch.nn as nn

from climategan_amd import tutils
from climategan_amd.config import default_opts
from oracle import ref_shim


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(48, 96, 3)
        self.bn = nn.BatchNorm2d(96)
        self.fc = nn.Linear(64, 32)
        self.plain_bn = nn.BatchNorm2d(8, affine=False)
        self.inorm = nn.InstanceNorm2d(96, affine=True)       # "InstanceNorm2d": neither rule matches


@pytest.mark.parametrize("init_type,gain", [("normal", 0.02), ("xavier", 0.02), ("xavier_uniform", 0.02),
                                            ("kaiming", 0.02), ("orthogonal", 0.5), ("none", 0.02)])
def test_init_types_follow_the_formulas(init_type, gain):
    torch.manual_seed(3)
    net = _Net()
    nn.init.constant_(net.conv.bias, 0.3)
    nn.init.constant_(net.inorm.weight, 0.7)
    tutils.init_weights(net, init_type, gain)
    fan_in, fan_out = 48 * 9, 96 * 9
    std = net.conv.weight.std().item()
    expect = {"normal": gain, "xavier": gain * math.sqrt(2.0 / (fan_in + fan_out)),
              "xavier_uniform": math.sqrt(2.0 / (fan_in + fan_out)),     # gain 1.0 whatever init_gain says
              "kaiming": math.sqrt(2.0 / fan_in),
              "orthogonal": gain / math.sqrt(fan_in),     # 96 orthonormal rows of length 432, times gain
              "none": 1.0 / math.sqrt(3 * fan_in)}[init_type]            # torch default: U(+-1/sqrt(fan_in))
    assert abs(std / expect - 1) < 0.03, (std, expect)
    if init_type != "none":
        assert net.conv.bias.abs().max() == 0 and net.fc.bias.abs().max() == 0
    assert abs(net.bn.weight.mean().item() - 1) < 0.4 * gain and abs(net.bn.weight.std().item() / gain - 1) < 0.3
    assert net.bn.bias.abs().max() == 0
    assert (net.inorm.weight == 0.7).all()


def test_falsy_arguments_and_unknown_type(capsys):
    net = _Net()
    tutils.init_weights(net, None, 0)
    assert "defaulting to normal" in capsys.readouterr().out
    assert abs(net.fc.weight.std().item() / 0.02 - 1) < 0.1
    with pytest.raises(NotImplementedError):
        tutils.init_weights(net, "he-he", 0.02)


def test_spectral_norm_convs_are_skipped():
    from climategan_amd.blocks import Conv2dBlock
    torch.manual_seed(0)
    blk = Conv2dBlock(8, 8, 3, padding=1, norm="spectral", pad_type="reflect")
    before = {k: v.clone() for k, v in blk.state_dict().items()}
    tutils.init_weights(blk, "xavier", 0.02)
    for k, v in blk.state_dict().items():
        assert torch.equal(v, before[k]), k       # no ``weight`` attribute -> untouched, bias included


def test_helpers():
    net = _Net()
    assert tutils.get_num_params(net) == sum(p.numel() for p in net.parameters())
    for p in net.parameters():
        p.grad = torch.ones_like(p)
    tutils.zero_grad(net)
    assert all(p.grad is None for p in net.parameters())
    t = torch.arange(24, dtype=torch.float32).reshape(2, 1, 3, 4) * torch.tensor([1.0, -2.0]).reshape(2, 1, 1, 1)
    n = tutils.normalize(t)
    assert n.reshape(2, -1).min(1)[0].tolist() == [0, 0] and n.reshape(2, -1).max(1)[0].tolist() == [1, 1]
    assert tutils.normalize(t[0], 2, 4).min() == 2 and tutils.normalize(t[0], 2, 4).max() == 4
    x = torch.tensor([-1.0, 0.0, 1.0]).reshape(1, 3, 1, 1)       # R, G, B
    v = tutils.vgg_preprocess(x).flatten().tolist()             # B, G, R in [0, 255] minus the caffe means
    assert v == pytest.approx([255 - 103.939, 127.5 - 116.779, 0 - 123.680])


# ------------------------------------------------------------------------------------- against the real reference
_RECORDED = ("normal_", "xavier_normal_", "xavier_uniform_", "kaiming_normal_", "orthogonal_", "constant_", "ones_",
             "zeros_", "kaiming_uniform_", "uniform_")


class _InitRecorder:
    """Wraps torch.nn.init's in-place initialisers and logs (function, storage pointer) per call."""

    def __enter__(self):
        self.calls = []
        self.saved = {n: getattr(nn.init, n) for n in _RECORDED}
        for n, f in self.saved.items():
            def wrapped(t, *a, _f=f, _n=n, **k):
                self.calls.append((_n, t.data_ptr(), tuple(round(float(x), 6) for x in a if isinstance(x, (int, float)))
                                   + tuple(sorted((kk, str(vv)) for kk, vv in k.items()))))
                return _f(t, *a, **k)
            setattr(nn.init, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(nn.init, n, f)

    def by_key(self, module):
        ptr2key = {}
        for k, v in module.state_dict(keep_vars=True).items():
            ptr2key.setdefault(v.data_ptr(), k)
        out = {}
        for n, ptr, args in self.calls:
            if ptr in ptr2key:
                out.setdefault(ptr2key[ptr], []).append((n,) + args)
        return out


def _opts_pair(tmp_path, ckpt):
    ro = ref_shim.default_opts()
    mo = default_opts()
    for o in (ro, mo):
        o.tasks = ["d", "s", "m", "p"]
        o.gen.p.latent_dim = 16
        o.gen.p.spade_n_up = 4
        o.dis.p.ndf = 8
        o.gen.deeplabv3.pretrained_model = {"resnet": str(ckpt), "mobilenet": str(ckpt)}
    return ro, mo


@pytest.mark.skipif(not ref_shim.available(), reason="/root/reference not present")
@pytest.mark.parametrize("use_spade,dis_gan", [(False, "WGAN_norm"), (True, "GAN")])
def test_same_init_calls_as_the_reference(tmp_path, use_spade, dis_gan):
    rgen = ref_shim.ref("generator")
    rdis = ref_shim.ref("discriminator")
    from climategan_amd.discriminator import create_discriminator
    from climategan_amd.generator import create_generator

    ckpt = tmp_path / "deeplab.pth"
    ro, mo = _opts_pair(tmp_path, ckpt)
    for o in (ro, mo):
        o.gen.m.use_spade = use_spade
        o.dis.m.gan_type = dis_gan
        o.dis.s.gan_type = dis_gan
    saved_cuda = nn.Module.cuda
    nn.Module.cuda = lambda self, *a, **k: self            # MaskSpadeDecoder hard-codes .cuda() (masker.py:196)
    try:
        # a stand-in for the pretrained DeepLab-v3+ checkpoint: backbone.* / aspp.* / decoder.* incl. a 19-class head
        torch.manual_seed(1)
        G0 = rgen.create_generator(ro, "cpu", no_init=True)
        std = {"backbone." + k: v for k, v in G0.encoder.state_dict().items()}
        std.update({"aspp." + k: v + 0.5 for k, v in G0.decoders["s"].aspp.state_dict().items()})
        std.update({"decoder." + k: v + 0.25 for k, v in G0.decoders["s"].decoder.state_dict().items()})
        std["decoder.conv_out.weight"] = torch.zeros(19, 256, 1, 1)           # Cityscapes head: must be filtered out
        torch.save(std, ckpt)
        with _InitRecorder() as rr:
            Gr = rgen.create_generator(ro, "cpu", no_init=False)
            Dr = rdis.create_discriminator(ro, "cpu", no_init=False)
        ref_g, ref_d = rr.by_key(Gr), rr.by_key(Dr)
    finally:
        nn.Module.cuda = saved_cuda
    with _InitRecorder() as mr:
        Gm = create_generator(mo, "cpu", no_init=False)
        Dm = create_discriminator(mo, "cpu", no_init=False)
    got_g, got_d = mr.by_key(Gm), mr.by_key(Dm)
    assert set(Gr.state_dict()) == set(Gm.state_dict()) and set(Dr.state_dict()) == set(Dm.state_dict())
    for ref, got, what in ((ref_g, got_g, "G"), (ref_d, got_d, "D")):
        assert set(ref) == set(got), (what, set(ref) ^ set(got))
        for k in ref:
            assert ref[k] == got[k], (what, k, ref[k], got[k])
    # the loaded checkpoint arrived in the same places (and the 19-class head nowhere)
    sr, sm = Gr.state_dict(), Gm.state_dict()
    for k in ("encoder.layer3.5.conv2.weight", "decoders.s.aspp.conv2.conv.weight", "decoders.s.decoder.conv_low.bn.bias"):
        assert torch.equal(sr[k], sm[k]), k
    assert sm["decoders.s.decoder.conv_out.weight"].shape[0] == 11
    # spot checks of the resulting distributions: xavier gain 0.02 on the depth head, N(1, 0.02) BatchNorm weights
    w = sm["decoders.d.enc4_2.conv.weight"]
    assert abs(w.std().item() / (0.02 * math.sqrt(2.0 / (512 * 9 + 512 * 9))) - 1) < 0.02
    assert abs(sm["decoders.d.enc4_2.norm.weight"].std().item() / 0.02 - 1) < 0.2
    assert sm["decoders.d.upsample.2.bias"].abs().max() == 0
    # no_init leaves everything at the constructors' draws: nothing from the xavier family is called
    with _InitRecorder() as nr:
        Gn = create_generator(mo, "cpu", no_init=True)
        Dn = create_discriminator(mo, "cpu", no_init=True)
    called = {c[0] for calls in list(nr.by_key(Gn).values()) + list(nr.by_key(Dn).values()) for c in calls}
    assert "xavier_normal_" not in called and "normal_" not in called


def test_missing_pretrained_checkpoint_asserts_like_the_reference(tmp_path):
    from climategan_amd.deeplab import build_v3_backbone
    o = default_opts()
    o.gen.deeplabv3.pretrained_model = {"resnet": str(tmp_path / "absent.pth")}
    with pytest.raises(AssertionError):
        build_v3_backbone(o, no_init=False)
    build_v3_backbone(o, no_init=True)                          # never looks at the path
    o.gen.deeplabv3.pretrained_model = {"resnet": "none"}
    build_v3_backbone(o, no_init=False)                         # explicit opt-out
