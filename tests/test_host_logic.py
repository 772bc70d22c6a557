"""CPU-side tests: the C-ABI library builds/loads and exports every declared symbol; host-side module
mirrors keep the reference's state-dict layout and option handling.  No GPU compute here."""
import ctypes
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from climategan_amd import _lib

    if not _lib.LIB_PATH.exists():
        import __graft_entry__ as g

        g.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    from climategan_amd import _lib

    header = (ROOT / "include" / "climategan_hip.h").read_text()
    declared = set(re.findall(r"\b(cgan_[a-z0-9_]+)\s*\(", header)) - {"cgan_cs", "cgan_cond_cs"}
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    raw = ctypes.CDLL(str(_lib.LIB_PATH))
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.cgan_version() == 1


def test_host_side_validation_without_gpu(lib):
    """Descriptor validation runs on the host before any launch: bad arguments return an error code + message."""
    from climategan_amd._lib import ConvDesc, SpadeDesc

    d = ConvDesc(0, 1, 8, 8, 8, 8, 3, 3, 1, 1, 1, 0, 9, 9, 0, 0, 0.2, 1, 0, 0)  # wrong h_out
    assert lib.cgan_conv2d_packed_weight_bytes(ctypes.byref(d)) == 0
    assert b"inconsistent" in lib.cgan_last_error()
    d = ConvDesc(0, 1, 8, 8, 20, 20, 3, 3, 1, 1, 1, 0, 8, 8, 0, 0, 0.2, 1, 0, 0)
    # 20 -> 24 storage channels, padded to 32 per tap for 3x3 kernels: K = 9*32 -> 9 k-steps; 24 rows -> 2 cout tiles
    assert lib.cgan_conv2d_packed_weight_bytes(ctypes.byref(d)) == 2 * 9 * 64 * 16
    d1 = ConvDesc(0, 1, 8, 8, 20, 20, 1, 1, 1, 0, 1, 0, 8, 8, 0, 0, 0.2, 1, 0, 0)
    assert lib.cgan_conv2d_packed_weight_bytes(ctypes.byref(d1)) == 2 * 1 * 64 * 16   # 1x1: dense K = 24 -> 1 k-step
    s = SpadeDesc(0, 1, 16, 16, 40, 0, 64, 64, 3, 64, 3, 0, 0.2)  # hidden != 128
    assert lib.cgan_spade_packed_weight_bytes(ctypes.byref(s)) == 0
    assert b"hidden must be 128" in lib.cgan_last_error()
    assert lib.cgan_spectral_norm_workspace_bytes(640, 5760) == (20 * 5760 + 5760 + 640 + 4) * 4


def test_rccl_entry_points_refuse_before_load(lib):
    """The gradient-bucket collective of the C ABI loads RCCL at run time: before cgan_rccl_load has succeeded every entry
    point fails with a message (no crash, no link-time dependency: this test runs on a box without a GPU), and a bad path
    is an error."""
    if lib.cgan_rccl_loaded():
        pytest.skip("RCCL already loaded in this process")
    buf = (ctypes.c_float * 4)()
    comm = ctypes.c_void_p()
    uid = ctypes.create_string_buffer(128)
    assert lib.cgan_allreduce_bucket(buf, 4, 2, ctypes.c_void_p(1), None) != 0
    assert b"cgan_rccl_load" in lib.cgan_last_error()
    assert lib.cgan_comm_unique_id(uid) != 0
    assert lib.cgan_comm_init_rank(ctypes.byref(comm), 1, uid, 0) != 0
    assert lib.cgan_rccl_load(b"/nonexistent/librccl.so") != 0
    assert b"dlopen" in lib.cgan_last_error()
    assert lib.cgan_rccl_loaded() == 0


def test_product_path_has_no_cpu_fallback():
    from climategan_amd import ops

    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.nchw_to_nhwc(torch.zeros(1, 3, 8, 8), torch.float16)


def test_product_never_imports_oracle():
    for f in (ROOT / "climategan_amd").rglob("*.py"):
        src = f.read_text()
        assert "import oracle" not in src and "from oracle" not in src, f


def test_painter_state_dict_layout():
    from climategan_amd.config import default_opts
    from climategan_amd.generator import create_generator
    from helpers import painter_shapes

    opts = default_opts()
    opts.tasks = ["p"]
    G = create_generator(opts)
    sd = G.painter.state_dict()
    assert len(sd) == 229  # SURVEY 8b [probe]
    assert {k: tuple(v.shape) for k, v in sd.items()} == painter_shapes(640, 7)
    uv = [k for k, p in G.painter.named_parameters() if k.endswith(("weight_u", "weight_v"))]
    assert len(uv) == 46 and all(not dict(G.painter.named_parameters())[k].requires_grad for k in uv)
    assert sum(p.numel() for p in G.painter.parameters()) == 42341735
    G.painter.set_latent_shape((8, 3, 640, 640), True)
    assert (G.painter.z_h, G.painter.z_w) == (5, 5)
    G.painter.set_latent_shape(7, False)
    assert (G.painter.z_h, G.painter.z_w) == (7, 7)


def test_unsupported_options_raise():
    from climategan_amd.blocks import SPADEResnetBlock
    from climategan_amd.config import default_opts
    from climategan_amd.norms import SPADE
    from climategan_amd.painter import PainterSpadeDecoder

    with pytest.raises(ValueError, match="not a recognized param-free norm"):
        SPADE("layer", 3, 8, 3)
    opts = default_opts()
    opts.gen.p.use_final_shortcut = True
    with pytest.raises(NotImplementedError):
        PainterSpadeDecoder(opts)
    assert SPADEResnetBlock(8, 4, 3, True, "instance", 3).learned_shortcut


def test_discriminator_state_dict_layout():
    from climategan_amd.config import default_opts
    from climategan_amd.discriminator import create_discriminator
    from helpers import disc_fc_shapes, disc_p_shapes

    D = create_discriminator(default_opts(), "cpu")
    sd = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    want = {"p." + k: v for k, v in disc_p_shapes(4, 64, 4, 3).items()}
    want.update({"m.Advent." + k: v for k, v in disc_fc_shapes(2).items()})
    want.update({"s.Advent." + k: v for k, v in disc_fc_shapes(11).items()})
    assert sd == want and len(sd) == 112
    n = sum(p.numel() for p in D.parameters())
    assert 26.4e6 < n < 26.6e6   # SURVEY section 6 [probe]: D_p 20.96 M + D_m 2.78 M + D_s 2.79 M


def test_generator_masker_state_dict_layout():
    """encoder / decoders.{d,s,m} keys and shapes equal the reference's default generator (fixture written by
    oracle/make_golden.py from the real reference); full G = 105.4 M parameters."""
    from climategan_amd.config import default_opts
    from climategan_amd.generator import create_generator
    from helpers import masker_shapes

    G = create_generator(default_opts())
    sd = {k: tuple(v.shape) for k, v in G.state_dict().items() if not k.startswith("painter.")}
    assert sd == masker_shapes()
    assert sum(p.numel() for p in G.parameters()) == 105414209


def test_pl4m_is_enabled_at_its_epoch():
    """``Trainer.maybe_enable_pl4m`` = the epoch check of the reference's ``train`` loop (trainer.py:899-909)."""
    from climategan_amd.config import default_opts
    from climategan_amd.trainer import Trainer

    class _P(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(n)) if n else None

    class _G:
        painter = _P(3)

    opts = default_opts()
    assert opts.gen.p.pl4m_epoch == 49 and opts.gen.m.use_pl4m is False          # defaults.yaml:152,176
    T = Trainer(opts, device="cpu")
    T.G, T.epoch = _G(), 0
    opts.gen.m.use_pl4m = True
    opts.gen.p.pl4m_epoch = 2
    assert T.maybe_enable_pl4m() is False
    T.epoch = 2
    assert T.maybe_enable_pl4m() is True and T.use_pl4m
    T.epoch = 3
    assert T.maybe_enable_pl4m() is True                                          # stays on
    T2 = Trainer(opts, device="cpu")
    T2.G, T2.epoch = _G(), 2
    opts.gen.m.use_pl4m = False
    assert T2.maybe_enable_pl4m() is False
    opts.gen.m.use_pl4m = True
    T2.G = type("G0", (), {"painter": _P(0)})()
    assert T2.maybe_enable_pl4m() is False                                        # no Painter of its own


def test_product_library_exports_no_development_knob():
    """libcgan_hip.so exports the declared ABI and nothing else of ours: the cgan_debug_* knobs live in the development build
    only (-DCGAN_DEV -> libcgan_hip_dev.so), which exports the same ABI plus the knobs."""
    import subprocess
    from climategan_amd import _lib

    if not _lib.LIB_PATH.exists() or not _lib.DEV_LIB_PATH.exists():
        import __graft_entry__ as g
        g.build()

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", str(path)], check=True, capture_output=True, text=True).stdout
        names = {line.split()[-1] for line in out.splitlines() if line.split()}
        # the dynamic symbol table holds the C ABI only (csrc/exports.map): no C++-mangled internals of the kernels' launchers
        foreign = sorted(n for n in names if not n.startswith("cgan_"))
        assert not foreign, "%s exports symbols outside the C ABI: %s" % (path.name, foreign[:8])
        return names

    prod, dev = exported(_lib.LIB_PATH), exported(_lib.DEV_LIB_PATH)
    assert prod == set(_lib.EXPORTED_SYMBOLS), sorted(prod ^ set(_lib.EXPORTED_SYMBOLS))
    assert not [s for s in prod if "debug" in s]
    knobs = dev - prod
    assert knobs and all(s.startswith("cgan_debug_") for s in knobs), sorted(knobs)
    assert prod <= dev


def test_float_cast_is_guarded_and_reversible():
    """``G.float()`` (the reference's fp32 apply_events run, apply_events.py:465-468) selects the split-precision Masker only
    where every module can run it, never in training mode, and ``train()`` / a 16-bit cast leave the mode again (advisor,
    round 4: the mode was sticky and half-switched generators it could not serve)."""
    import pytest
    from climategan_amd.config import default_opts
    from climategan_amd.generator import create_generator

    opts = default_opts()
    opts.tasks = ["d", "s", "m"]
    G = create_generator(opts, no_init=True)
    G.half()
    assert G.train().float().pair_precision is False and G.compute_dtype == torch.float16      # training mode: no-op
    G.eval().float()
    assert G.pair_precision and G.encoder.pair_precision and G.compute_dtype == torch.bfloat16  # split24 = bf16 triples
    G.train()
    assert not G.pair_precision and not G.encoder.pair_precision and G.compute_dtype == torch.float16
    G.eval().float().bfloat16()
    assert not G.pair_precision and G.compute_dtype == torch.bfloat16

    opts2 = default_opts()
    opts2.tasks = ["d", "s", "m"]
    opts2.gen.m.use_spade = True
    G2 = create_generator(opts2, no_init=True)
    before = G2.compute_dtype
    G2.eval().float()
    assert G2.pair_precision                                  # round 5: the SPADE mask decoder runs on split maps too
    G2.set_compute_dtype(before)
    G2.encoder = torch.nn.Identity()                          # an encoder without a split-precision path
    G2.eval().float()
    assert not G2.pair_precision and G2.compute_dtype == before                   # unsupported: keeps the 16-bit type
    with pytest.raises(NotImplementedError):
        G2.set_compute_dtype("split24")


def test_chunk_arena_joins_batch_slices_without_a_copy():
    """ops._run_chunks: the slices of a batch-chunked op write consecutive ranges of ONE full-batch buffer (no torch.cat);
    anything the arena cannot serve falls back to a concatenation with the same values."""
    import torch

    from climategan_amd import ops

    x = torch.arange(7 * 2 * 3 * 8, dtype=torch.float32).reshape(7, 2, 3, 8)

    def op(lo, cnt):                      # an op with a 4-D map, [N, Cs] statistics, a scratch buffer and a non-arena result
        xs = x[lo:lo + cnt]
        scratch = ops._empty(64, dtype=torch.uint8, device="cpu")        # 1-D: never from the arena
        y = ops._empty((cnt, 2, 3, 8), dtype=torch.float32, device="cpu")
        y.copy_(xs * 2)
        mean = ops._empty((cnt, 8), dtype=torch.float32, device="cpu")
        mean.copy_(xs.mean((1, 2)))
        z = ops._empty_like(y)
        z.copy_(xs + 1)
        assert scratch.numel() == 64
        return ops.NHWC(y, 8), (mean, z.clone())                         # the clone is NOT an arena range -> joined by cat

    out, (mean, z) = ops._run_chunks(7, 3, op)                           # slices of 3, 3, 1
    assert ops._ARENA is None
    assert torch.equal(out.t, x * 2) and torch.equal(mean, x.mean((1, 2))) and torch.equal(z, x + 1)
    assert out.t.shape[0] == 7 and out.t.is_contiguous()
    # the map and the statistics came out of the arena's full-batch buffers: nothing was concatenated
    arena_ptrs = set()

    def op2(lo, cnt):
        y = ops._empty((cnt, 2, 3, 8), dtype=torch.float32, device="cpu")
        arena_ptrs.add(ops._ARENA.full[0].data_ptr())
        y.fill_(float(lo))
        return y

    y = ops._run_chunks(7, 3, op2)
    assert y.data_ptr() in arena_ptrs and y[:, 0, 0, 0].tolist() == [0, 0, 0, 3, 3, 3, 6]
    # an allocation sequence that differs between slices falls back to cat instead of handing out a wrong buffer
    def op3(lo, cnt):
        if lo == 0:
            ops._empty((cnt, 1, 1, 8), dtype=torch.float32, device="cpu")
        y = ops._empty((cnt, 2, 3, 8), dtype=torch.float32, device="cpu")
        y.fill_(float(lo))
        return y

    y = ops._run_chunks(7, 3, op3)
    assert y[:, 0, 0, 0].tolist() == [0, 0, 0, 3, 3, 3, 6]
