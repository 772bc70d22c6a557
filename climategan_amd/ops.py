"""Thin Python wrappers over the C ABI (include/climategan_hip.h): tensors in, raw device pointers out.

PyTorch is used only as the device-memory container / stream provider (tensor.data_ptr(),
torch.cuda.current_stream()).  Every op here runs a hand-written HIP kernel from libcgan_hip.so; there is no
torch fallback.
"""
import ctypes as C
import functools
import inspect
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import (ACT_LRELU, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, CGAN_BF16, CGAN_F16, PAD_REFLECT,  # noqa: F401
                   PAD_ZERO, ConvDesc, NormStatsDesc, PackItem, SnItem, SpadeDesc)

_DT = {torch.float16: CGAN_F16, torch.bfloat16: CGAN_BF16}


def touch(*tensors) -> None:
    """Tell torch that these tensors were modified in place.  The HIP kernels write module state (parameters in the
    optimizer, BatchNorm running statistics) through raw pointers; the packed-weight caches (norms._PackCache) key on
    ``tensor._version``, so every such write must bump it -- without this the forward keeps using the weights packed
    before the first optimizer step.  No kernel launch.  Pass the module's own tensor objects, not ``.data`` aliases
    (those carry their own counter)."""
    ts = [t for t in tensors if t is not None]
    if ts:
        torch.autograd.graph.increment_version(ts)


MAX_MAP_BYTES = 2 ** 31      # the sample-independent ops slice the batch from here on (_batch_chunked; tests lower it)
ABI_MAX_BYTES = 2 ** 31      # nothing this large is ever handed to a C-ABI call (_ptr)


def cs8(c: int) -> int:
    return (c + 7) & ~7


def cs4(c: int) -> int:
    return (c + 3) & ~3


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current stream of the current device, as the C ABI's ``void* stream``.  Every kernel launch asks for it:
    the raw getters cost ~0.3 us, ``torch.cuda.current_stream().cuda_stream`` ~4 us (a Stream object per call) -- 4 300
    launches per train step from one host thread (no measurable change of the step: the device, not the host, is the
    limit; it is simply less work)."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return C.c_void_p(_RAW_STREAM(_GET_DEVICE()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return C.c_void_p(0)
    nb = t.nbytes
    if nb >= ABI_MAX_BYTES:
        # several kernels address a tensor with 32-bit byte offsets (buffer descriptors, LDS-DMA lane offsets): nothing of
        # 2 GiB or more is ever handed to the C ABI.  The sample-independent ops split the batch themselves
        # (``_batch_chunked``); an op that reaches this line has no such split
        raise RuntimeError("climategan_amd.ops: a %s %s tensor of %.2f GiB exceeds the 2 GiB one C-ABI call covers (32-bit "
                           "offsets in the kernels) and this op does not split the batch; use a smaller batch per call"
                           % (tuple(t.shape), t.dtype, nb / 2 ** 30))
    if _lib.CALL_LOG is not None:          # development aid, see _lib.CALL_LOG
        _lib.log_bytes(nb)
    return C.c_void_p(t.data_ptr())


def _need_cs8(what, *xs):
    """The conv / norm kernels read round_up(c, 8) storage channels per pixel; only the conditioning image of the SPADE
    kernels is stored with round_up(c, 4)."""
    for x in xs:
        if x is not None and x.cs != cs8(x.c):
            raise RuntimeError("%s: tensor with %d logical channels must be stored with %d channels, got %d"
                               % (what, x.c, cs8(x.c), x.cs))


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("climategan_amd ops need device tensors (got a %s tensor); there is no CPU path"
                               % t.device)
    # These functions launch kernels and record nothing for autograd.  Inside an ``autograd.Function`` (forward and
    # backward run with grad mode off) that is the point; anywhere else a tensor that carries a graph would come out
    # silently detached -- refuse, the caller wants the grad-aware wrapper (``functional`` / ``autograd``).
    if torch.is_grad_enabled():
        for t in ts:
            if t is not None and t.requires_grad and not t.is_leaf:
                raise RuntimeError("climategan_amd.ops: called with grad mode on and a tensor that carries an autograd "
                                   "graph: the result would be silently detached; use climategan_amd.functional / "
                                   "autograd (HIP backward) or detach() / no_grad() explicitly")


def detached(x: "NHWC") -> "NHWC":
    """The same map without its autograd graph (an explicit stop-gradient for the ops that need one)."""
    return x.detach() if isinstance(x, PairMap) else NHWC(x.t.detach(), x.c)


@dataclass
class NHWC:
    """Channel-padded NHWC 16-bit activation: ``t`` is [N,H,W,Cs] (Cs = storage channels), ``c`` logical."""
    t: torch.Tensor
    c: int

    def __post_init__(self):
        # the kernels read cs8(c) (conditioning maps: cs4(c)) storage channels per pixel: anything else is out of bounds
        if self.t.dim() != 4 or self.t.shape[3] not in (cs8(self.c), cs4(self.c)):
            raise RuntimeError("NHWC: a [N,H,W,Cs] tensor with Cs = %d (or %d) storage channels is needed for %d logical "
                               "channels, got shape %s" % (cs8(self.c), cs4(self.c), self.c, tuple(self.t.shape)))

    @property
    def n(self): return self.t.shape[0]
    @property
    def h(self): return self.t.shape[1]
    @property
    def w(self): return self.t.shape[2]
    @property
    def cs(self): return self.t.shape[3]
    @property
    def dtype_id(self): return _DT[self.t.dtype]
    @property
    def shape(self):
        """The logical NCHW shape: what the reference's callers read off a latent (``z[0].shape[0]``, trainer.py:602-607)."""
        return torch.Size((self.t.shape[0], self.c, self.t.shape[1], self.t.shape[2]))

    def detach(self) -> "NHWC":
        return NHWC(self.t.detach(), self.c)


def split_blocks(dtype) -> int:
    """K blocks a split-precision conv multiplies per pixel (csrc/cgan_common.h, Split<T>::NB): fp16 pairs (hi | lo | hi) = 3,
    bf16 triples (hi | mid | lo | hi | mid | hi) = 6 -- the input-channel extent of the expanded weights."""
    return 6 if dtype == torch.bfloat16 else 3


def store_blocks(dtype) -> int:
    """Channel blocks per pixel a split-precision map STORES (Split<T>::NS, round 6): every component once -- fp16 (hi | lo) =
    2, bf16 (hi | mid | lo) = 3; the conv kernels read K block b from storage block xcomp(b)."""
    return 3 if dtype == torch.bfloat16 else 2


@dataclass
class PairMap:
    """Split-precision activation map of the inference-time Masker (``G.float()`` / ``set_compute_dtype("split24" |
    "pair16")``, csrc/pair.hip): every value is carried as several 16-bit numbers whose sum it is -- fp16: hi + lo, bf16:
    hi + mid + lo; ``t`` is [N,H,W,NB*Cs] with NB = ``split_blocks`` channel blocks, Cs = round_up(c, 8).  ``sigmoid``: a sigmoid still to be applied when the map leaves as fp32 NCHW (generator.py:277)."""
    t: torch.Tensor
    c: int
    sigmoid: bool = False

    def __post_init__(self):
        if self.t.dim() != 4 or self.t.shape[3] != store_blocks(self.t.dtype) * cs8(self.c):
            raise RuntimeError("PairMap: a [N,H,W,%d*%d] tensor is needed for %d logical channels, got shape %s"
                               % (store_blocks(self.t.dtype), cs8(self.c), self.c, tuple(self.t.shape)))

    @property
    def n(self): return self.t.shape[0]
    @property
    def h(self): return self.t.shape[1]
    @property
    def w(self): return self.t.shape[2]
    @property
    def cs(self): return self.t.shape[3] // store_blocks(self.t.dtype)
    @property
    def nb(self): return store_blocks(self.t.dtype)          # stored blocks per pixel
    @property
    def kb(self): return split_blocks(self.t.dtype)          # K blocks a conv multiplies
    @property
    def dtype_id(self): return _DT[self.t.dtype]
    @property
    def shape(self):
        return torch.Size((self.t.shape[0], self.c, self.t.shape[1], self.t.shape[2]))

    def detach(self) -> "PairMap":
        return PairMap(self.t.detach(), self.c, self.sigmoid)


def _plain_pair(x: "PairMap", what: str):
    if x.sigmoid:
        raise RuntimeError("%s: a pair map with a pending sigmoid can only leave through nhwc_to_nchw" % what)
    _need_cuda(x.t)


def pair_from_nchw(x: torch.Tensor, dtype: torch.dtype) -> PairMap:
    """fp32 NCHW -> (hi | lo | hi) pair map."""
    _need_cuda(x)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    y = _empty((n, h, w, store_blocks(dtype) * cs8(c)), dtype=dtype, device=x.device)
    _lib.check(_lib.load().cgan_pair_from_nchw(_ptr(x), _ptr(y), _DT[dtype], n, c, h, w, _stream()), "cgan_pair_from_nchw")
    return PairMap(y, c)


def pair_to_nhwc(x: PairMap) -> NHWC:
    """hi + lo rounded once to the 16-bit type: the ordinary map the event kernels and the Painter read."""
    _plain_pair(x, "pair_to_nhwc")
    y = _empty((x.n, x.h, x.w, x.cs), dtype=x.t.dtype, device=x.t.device)
    _lib.check(_lib.load().cgan_pair_to_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n * x.h * x.w, x.c, _stream()),
               "cgan_pair_to_nhwc")
    return NHWC(y, x.c)


# ------------------------------------------------------------------------------------------------ maps of 2 GiB and more
class _ChunkArena:
    """Result buffers of an op that runs on batch slices (round 6).  NHWC keeps a sample contiguous, so the slices' results
    are consecutive ranges of ONE full-batch tensor: while an arena is active, ``_empty`` serves every [count, ...] map the
    op allocates (4-D maps and [N, Cs] statistics, in allocation order) as the slice's range of a full-batch buffer
    allocated on the first slice, and ``_join`` hands that buffer out as the result -- no concatenation pass (rounds 4-5
    joined the slices with ``torch.cat``: 14 extra passes over 2-3.4 GB maps per train step at 32 samples per domain,
    17.9 ms of a 634 ms step)."""

    def __init__(self, n: int):
        self.n, self.full, self.j, self.lo, self.cnt = n, [], 0, 0, 0

    def begin(self, lo: int, cnt: int) -> None:
        self.lo, self.cnt, self.j = lo, cnt, 0

    def take(self, shape, dtype, device):
        if len(shape) not in (2, 4) or shape[0] != self.cnt:
            return None
        j = self.j
        if j == len(self.full):
            if self.lo != 0:          # a later slice allocates a map the first one did not: plain allocation (joined by cat)
                return None
            self.full.append(torch.empty((self.n,) + tuple(shape[1:]), dtype=dtype, device=device))
        f = self.full[j]
        if tuple(f.shape[1:]) != tuple(shape[1:]) or f.dtype != dtype:
            return None               # (the slices' allocation sequences differ: plain allocation, joined by cat)
        self.j = j + 1
        return f[self.lo:self.lo + self.cnt]

    def whole(self, pieces, spans):
        """The full-batch buffer whose ranges ``pieces`` (one tensor per slice) are, or None."""
        t0 = pieces[0]
        for f in self.full:
            if f.data_ptr() == t0.data_ptr() and f.dtype == t0.dtype and tuple(f.shape[1:]) == tuple(t0.shape[1:]):
                if all(t.is_contiguous() and t.shape[0] == cnt and t.data_ptr() == f[lo:lo + cnt].data_ptr()
                       for t, (lo, cnt) in zip(pieces, spans)):
                    return f
        return None


_ARENA = None


def _empty(shape, dtype=None, device=None):
    if _ARENA is not None and not isinstance(shape, int):
        t = _ARENA.take(tuple(shape), dtype if dtype is not None else torch.get_default_dtype(), device)
        if t is not None:
            return t
    return torch.empty(shape, dtype=dtype, device=device)


def _empty_like(t: torch.Tensor):
    if _ARENA is not None and t.is_contiguous():
        r = _ARENA.take(tuple(t.shape), t.dtype, t.device)
        if r is not None:
            return r
    return torch.empty_like(t)


def _join(outs, spans, arena):
    """Per-slice results -> the full-batch result: the arena's buffer where the slices wrote into one, else a concatenation
    along the batch.  NHWC maps, [N, ...] tensors, tuples of those (None stays None)."""
    first = outs[0]
    if first is None:
        return None
    if isinstance(first, tuple):
        return tuple(_join([o[i] for o in outs], spans, arena) for i in range(len(first)))
    if isinstance(first, (NHWC, PairMap)):
        f = arena.whole([o.t for o in outs], spans)
        t = f if f is not None else torch.cat([o.t for o in outs], 0)
        return NHWC(t, first.c) if isinstance(first, NHWC) else PairMap(t, first.c, first.sigmoid)
    if isinstance(first, torch.Tensor):
        f = arena.whole(outs, spans)
        return f if f is not None else torch.cat(outs, 0)
    raise TypeError("cannot join chunk results of type %s" % type(first))


def _run_chunks(n: int, k: int, call):
    """``call(lo, count)`` over the batch in slices of ``k`` samples, results joined (see _ChunkArena)."""
    global _ARENA
    prev, arena = _ARENA, _ChunkArena(n)
    outs, spans = [], []
    try:
        _ARENA = arena
        for lo in range(0, n, k):
            cnt = min(k, n - lo)
            arena.begin(lo, cnt)
            outs.append(call(lo, cnt))
            spans.append((lo, cnt))
    finally:
        _ARENA = prev
    return _join(outs, spans, arena)


def _batch_chunked(*split, out_bytes_per_sample=None):
    """For ops whose samples are independent (NHWC keeps a sample contiguous: a batch slice is a view).  One C-ABI call
    covers maps below MAX_MAP_BYTES (32-bit offsets in the kernels); when an operand named in ``split`` (NHWC maps or
    [N, ...] tensors) or the result (``out_bytes_per_sample(get)`` with ``get(name)`` -> the call's argument) would reach
    that, the op runs on batch slices that write consecutive ranges of one full-batch result (_ChunkArena; only in that
    regime): configs[3]'s global batch of 32 per domain on ONE GPU has 3.4 GB maps (the SPADE hidden map, VGG's first
    block).  Below 1/256 of the limit nothing is examined (no op grows a map 256-fold)."""
    def deco(fn):
        params = list(inspect.signature(fn).parameters)
        defaults = {k: v.default for k, v in inspect.signature(fn).parameters.items()}
        idx = [(params.index(name), name) for name in split]

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            big = 0
            for v in args:
                if type(v) is NHWC or type(v) is PairMap:
                    big = max(big, v.t.nbytes)
                elif type(v) is torch.Tensor:
                    big = max(big, v.nbytes)
            for v in kwargs.values():
                if type(v) is NHWC or type(v) is PairMap:
                    big = max(big, v.t.nbytes)
                elif type(v) is torch.Tensor:
                    big = max(big, v.nbytes)
            if big * 256 < MAX_MAP_BYTES:
                return fn(*args, **kwargs)

            def get(name):
                i = params.index(name)
                return args[i] if i < len(args) else kwargs.get(name, defaults[name])

            n, per_sample = None, 0
            for i, name in idx:
                v = args[i] if i < len(args) else kwargs.get(name)
                if v is None:
                    continue
                t = v.t if type(v) in (NHWC, PairMap) else v
                n = t.shape[0]
                per_sample = max(per_sample, t.nbytes // max(n, 1))
            if n is None:
                return fn(*args, **kwargs)
            if out_bytes_per_sample is not None:
                per_sample = max(per_sample, int(out_bytes_per_sample(get)))
            if n * per_sample < MAX_MAP_BYTES:
                return fn(*args, **kwargs)
            k = (MAX_MAP_BYTES - 1) // per_sample
            if k < 1:
                raise RuntimeError("%s: ONE sample's map (%.2f GiB) exceeds the 2 GiB a C-ABI call covers"
                                   % (fn.__name__, per_sample / 2 ** 30))

            def call(lo, cnt):
                a, kw = list(args), dict(kwargs)
                for i, name in idx:
                    v = a[i] if i < len(a) else kw.get(name)
                    if v is None:
                        continue
                    piece = (NHWC(v.t[lo:lo + cnt], v.c) if type(v) is NHWC else
                             PairMap(v.t[lo:lo + cnt], v.c, v.sigmoid) if type(v) is PairMap else v[lo:lo + cnt])
                    if i < len(a):
                        a[i] = piece
                    else:
                        kw[name] = piece
                return fn(*a, **kw)
            return _run_chunks(n, k, call)
        return wrapper
    return deco


def _nbf(x) -> int:
    """channel blocks per pixel: 1 for an ordinary map, 3 / 6 for a split-precision one"""
    return x.nb if isinstance(x, PairMap) else 1


def _conv_out_hw(h, w, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1, (w + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ------------------------------------------------------------------------------------------------ layout
@_batch_chunked("x", "mask", out_bytes_per_sample=lambda g: g("x").shape[2] * g("x").shape[3] * (g("cs") or cs8(g("x").shape[1])) * 2)
def nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype, cs: Optional[int] = None,
                 mask: Optional[torch.Tensor] = None) -> NHWC:
    """fp32 NCHW -> 16-bit NHWC (optionally times (1 - mask): cond = x * (1 - m), reference generator.py:294)."""
    _need_cuda(x, mask)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    cs = cs or cs8(c)
    if mask is not None:
        mask = mask.contiguous().float()
        assert mask.shape == (n, 1, h, w), mask.shape
    y = _empty((n, h, w, cs), dtype=dtype, device=x.device)
    lib = _lib.load()
    _lib.check(lib.cgan_nchw_to_nhwc(_ptr(x), _ptr(mask), _ptr(y), _DT[dtype], n, c, h, w, cs, _stream()),
               "cgan_nchw_to_nhwc")
    return NHWC(y, c)


@_batch_chunked("y", "paste_x", "paste_m", out_bytes_per_sample=lambda g: g("y").h * g("y").w * g("y").c * 4)
def nhwc_to_nchw(y: NHWC, paste_x: Optional[torch.Tensor] = None, paste_m: Optional[torch.Tensor] = None):
    """16-bit NHWC -> fp32 NCHW; with paste: out = paste_x * (1 - m) + y * m (reference generator.py:295-296)."""
    if isinstance(y, PairMap):
        if paste_x is not None:
            raise RuntimeError("nhwc_to_nchw: the paste is not available on a pair map")
        _need_cuda(y.t)
        out = _empty((y.n, y.c, y.h, y.w), dtype=torch.float32, device=y.t.device)
        _lib.check(_lib.load().cgan_pair_to_nchw(_ptr(y.t), _ptr(out), y.dtype_id, y.n, y.c, y.h, y.w, int(y.sigmoid),
                                                 _stream()), "cgan_pair_to_nchw")
        return out
    _need_cuda(y.t, paste_x, paste_m)
    n, h, w, cs = y.t.shape
    out = _empty((n, y.c, h, w), dtype=torch.float32, device=y.t.device)
    if paste_x is not None:
        paste_x = paste_x.contiguous().float()
        paste_m = paste_m.contiguous().float()
        assert paste_x.shape == out.shape and paste_m.shape == (n, 1, h, w)
    lib = _lib.load()
    _lib.check(lib.cgan_nhwc_to_nchw(_ptr(y.t), _ptr(paste_x), _ptr(paste_m), _ptr(out), y.dtype_id, n, y.c, h, w, cs,
                                     _stream()), "cgan_nhwc_to_nchw")
    return out


@_batch_chunked("x", out_bytes_per_sample=lambda g: g("size")[0] * g("size")[1] * (g("cs_out") or g("x").cs) * 2 * _nbf(g("x")))
def resize_nearest(x: NHWC, size: Tuple[int, int], cs_out: Optional[int] = None) -> NHWC:
    if isinstance(x, PairMap):
        _plain_pair(x, "resize_nearest")
        y = _empty((x.n, size[0], size[1], x.nb * x.cs), dtype=x.t.dtype, device=x.t.device)
        _lib.check(_lib.load().cgan_pair_resize_nearest(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, size[0], size[1],
                                                        _stream()),
                   "cgan_pair_resize_nearest")
        return PairMap(y, x.c)
    _need_cuda(x.t)
    oh, ow = size
    cs_out = cs_out or x.cs
    y = _empty((x.n, oh, ow, cs_out), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_resize_nearest_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, x.cs, oh, ow, cs_out,
                                            _stream()), "cgan_resize_nearest_nhwc")
    return NHWC(y, x.c)


@_batch_chunked("dy", out_bytes_per_sample=lambda g: g("size_in")[0] * g("size_in")[1] * g("cs_in") * 2)
def resize_nearest_bwd(dy: NHWC, size_in: Tuple[int, int], cs_in: int) -> NHWC:
    """Adjoint of ``resize_nearest``: the gradient of the (h_in, w_in) source map with ``cs_in`` storage channels."""
    _need_cuda(dy.t)
    hi, wi = size_in
    dx = _empty((dy.n, hi, wi, cs_in), dtype=dy.t.dtype, device=dy.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_resize_nearest_bwd_nhwc(_ptr(dy.t), _ptr(dx), dy.dtype_id, dy.n, dy.c, hi, wi, cs_in, dy.h, dy.w,
                                                dy.cs, _stream()), "cgan_resize_nearest_bwd_nhwc")
    return NHWC(dx, dy.c)


@_batch_chunked("x")
def avgpool3x3s2(x: NHWC) -> NHWC:
    _need_cuda(x.t)
    oh, ow = (x.h + 2 - 3) // 2 + 1, (x.w + 2 - 3) // 2 + 1
    y = _empty((x.n, oh, ow, x.cs), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_avgpool3x3s2_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, _stream()),
               "cgan_avgpool3x3s2_nhwc")
    return NHWC(y, x.c)


@_batch_chunked("x")
def maxpool3x3s2(x: NHWC) -> NHWC:
    if isinstance(x, PairMap):
        _plain_pair(x, "maxpool3x3s2")
        y = _empty((x.n, (x.h + 2 - 3) // 2 + 1, (x.w + 2 - 3) // 2 + 1, x.nb * x.cs), dtype=x.t.dtype, device=x.t.device)
        _lib.check(_lib.load().cgan_pair_maxpool3x3s2(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, _stream()),
                   "cgan_pair_maxpool3x3s2")
        return PairMap(y, x.c)
    _need_cuda(x.t)
    oh, ow = (x.h + 2 - 3) // 2 + 1, (x.w + 2 - 3) // 2 + 1
    y = _empty((x.n, oh, ow, x.cs), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_maxpool3x3s2_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, _stream()),
               "cgan_maxpool3x3s2_nhwc")
    return NHWC(y, x.c)


@_batch_chunked("x", out_bytes_per_sample=lambda g: g("size")[0] * g("size")[1] * g("x").cs * 2 * _nbf(g("x")))
def resize_bilinear(x: NHWC, size: Tuple[int, int], align_corners: bool = False) -> NHWC:
    if isinstance(x, PairMap):
        _plain_pair(x, "resize_bilinear")
        y = _empty((x.n, size[0], size[1], x.nb * x.cs), dtype=x.t.dtype, device=x.t.device)
        _lib.check(_lib.load().cgan_pair_resize_bilinear(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, size[0], size[1],
                                                         int(bool(align_corners)), _stream()), "cgan_pair_resize_bilinear")
        return PairMap(y, x.c)
    _need_cuda(x.t)
    oh, ow = size
    if x.cs != cs8(x.c):
        raise RuntimeError("resize_bilinear: input must be stored with round_up(c, 8) channels")
    y = _empty((x.n, oh, ow, x.cs), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_resize_bilinear_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, oh, ow,
                                             int(bool(align_corners)), _stream()), "cgan_resize_bilinear_nhwc")
    return NHWC(y, x.c)


def resize_bicubic(x: NHWC, size: Tuple[int, int]) -> NHWC:
    """F.interpolate(mode="bicubic", align_corners=False) (reference depth.py:143-149)."""
    _need_cuda(x.t)
    h, w = int(size[0]), int(size[1])
    if isinstance(x, PairMap):
        _plain_pair(x, "resize_bicubic")
        y = _empty((x.n, h, w, x.nb * x.cs), dtype=x.t.dtype, device=x.t.device)
        _lib.check(_lib.load().cgan_pair_resize_bicubic(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, h, w, _stream()),
                   "cgan_pair_resize_bicubic")
        return PairMap(y, x.c)
    y = _empty((x.n, h, w, cs8(x.c)), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_resize_bicubic_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, h, w, _stream()),
               "cgan_resize_bicubic_nhwc")
    return NHWC(y, x.c)


def concat_channels(xs) -> NHWC:
    """torch.cat(xs, dim=1) on NHWC tensors; every input but the last must have a multiple-of-8 channel count."""
    n, h, w = xs[0].n, xs[0].h, xs[0].w
    c_total = sum(x.c for x in xs)
    if isinstance(xs[0], PairMap):
        y = torch.zeros((n, h, w, xs[0].nb * cs8(c_total)), dtype=xs[0].t.dtype, device=xs[0].t.device)
        off = 0
        for x in xs:
            if not isinstance(x, PairMap) or (x.n, x.h, x.w) != (n, h, w) or x.t.dtype != y.dtype:
                raise RuntimeError("concat_channels: pair maps of one shape / dtype only")
            _plain_pair(x, "concat_channels")
            _lib.check(_lib.load().cgan_pair_copy_channels(_ptr(x.t), _ptr(y), x.dtype_id, n * h * w, x.c, c_total, off, _stream()),
                       "cgan_pair_copy_channels")
            off += x.c
        return PairMap(y, c_total)
    y = torch.zeros((n, h, w, cs8(c_total)), dtype=xs[0].t.dtype, device=xs[0].t.device)
    lib = _lib.load()
    off = 0
    for i, x in enumerate(xs):
        _need_cuda(x.t)
        if (x.n, x.h, x.w) != (n, h, w) or x.t.dtype != y.dtype:
            raise RuntimeError("concat_channels: shape / dtype mismatch")
        if off % 8 != 0:
            raise RuntimeError("concat_channels: channel offset %d is not a multiple of 8" % off)
        _lib.check(lib.cgan_copy_channels_nhwc(_ptr(x.t), _ptr(y), n * h * w, x.c, x.cs, y.shape[3], off, _stream()),
                   "cgan_copy_channels_nhwc")
        off += x.c
    return NHWC(y, c_total)


@_batch_chunked("a", "b")
def eltwise_mul(a: NHWC, b: NHWC) -> NHWC:
    if isinstance(a, PairMap):
        if not isinstance(b, PairMap) or a.t.shape != b.t.shape or a.c != b.c:
            raise RuntimeError("eltwise_mul: two pair maps of one shape expected")
        _plain_pair(a, "eltwise_mul")
        _plain_pair(b, "eltwise_mul")
        y = _empty_like(a.t)
        _lib.check(_lib.load().cgan_pair_mul(_ptr(a.t), _ptr(b.t), _ptr(y), a.dtype_id, a.n * a.h * a.w, a.c, _stream()),
                   "cgan_pair_mul")
        return PairMap(y, a.c)
    _need_cuda(a.t, b.t)
    if a.t.shape != b.t.shape or a.c != b.c:
        raise RuntimeError("eltwise_mul: shape mismatch")
    y = _empty_like(a.t)
    lib = _lib.load()
    _lib.check(lib.cgan_eltwise_nhwc(_ptr(a.t), _ptr(b.t), _ptr(y), a.dtype_id, 0, a.t.numel(), _stream()),
               "cgan_eltwise_nhwc")
    return NHWC(y, a.c)


def sigmoid(a: NHWC) -> NHWC:
    """Elementwise sigmoid (pad channels are re-zeroed by the caller if it matters: sigmoid(0) = 0.5)."""
    if isinstance(a, PairMap):          # applied in fp32 when the map leaves as NCHW (cgan_pair_to_nchw)
        _plain_pair(a, "sigmoid")
        return PairMap(a.t, a.c, sigmoid=True)
    _need_cuda(a.t)
    y = _empty_like(a.t)
    lib = _lib.load()
    _lib.check(lib.cgan_eltwise_nhwc(_ptr(a.t), _ptr(None), _ptr(y), a.dtype_id, 1, a.t.numel(), _stream()),
               "cgan_eltwise_nhwc")
    return NHWC(y, a.c)


def scale_by_scalar(a: NHWC, s: torch.Tensor) -> NHWC:
    """a * s with s a device fp32 scalar tensor (no host sync)."""
    _need_cuda(a.t, s)
    if s.dtype != torch.float32 or s.numel() != 1:
        raise RuntimeError("scale_by_scalar: one fp32 device scalar expected")
    y = _empty_like(a.t)
    lib = _lib.load()
    _lib.check(lib.cgan_eltwise_nhwc(_ptr(a.t), _ptr(s), _ptr(y), a.dtype_id, 2, a.t.numel(), _stream()),
               "cgan_eltwise_nhwc")
    return NHWC(y, a.c)


def fold_bn(w: torch.Tensor, bias, bn_weight, bn_bias, running_mean, running_var, eps: float):
    """Eval-mode BatchNorm folded into the preceding conv: returns (w', b') fp32 device tensors."""
    _need_cuda(w, bias, bn_weight, bn_bias, running_mean, running_var)
    w = w.detach().contiguous().float()
    c_out = w.shape[0]
    w_out = _empty_like(w)
    b_out = _empty(c_out, dtype=torch.float32, device=w.device)
    ts = [t.detach().contiguous().float() if t is not None else None
          for t in (bias, bn_weight, bn_bias, running_mean, running_var)]
    lib = _lib.load()
    _lib.check(lib.cgan_fold_bn(_ptr(w), _ptr(ts[0]), _ptr(ts[1]), _ptr(ts[2]), _ptr(ts[3]), _ptr(ts[4]), float(eps),
                                _ptr(w_out), _ptr(b_out), c_out, w.numel() // c_out, _stream()), "cgan_fold_bn")
    return w_out, b_out


# ------------------------------------------------------------------------------------------------ output post-ops
def normalize_to_uint8(x: torch.Tensor) -> torch.Tensor:
    """Per-image min-max normalise an NCHW fp32/fp16 tensor and convert to uint8 NHWC
    (reference trainer.py:311-326 + tutils.normalize, tutils.py:567-576)."""
    _need_cuda(x)
    if x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("normalize_to_uint8: fp32 or fp16 input expected, got %s" % x.dtype)
    x = x.contiguous()
    n, c, h, w = x.shape
    lib = _lib.load()
    nbytes = lib.cgan_normalize_u8_workspace_bytes(n)
    ws = _empty(nbytes, dtype=torch.uint8, device=x.device)
    out = _empty((n, h, w, c), dtype=torch.uint8, device=x.device)
    _lib.check(lib.cgan_normalize_u8_nhwc(_ptr(x), int(x.dtype == torch.float16), _ptr(out), n, c, h, w, _ptr(ws),
                                          nbytes, _stream()), "cgan_normalize_u8_nhwc")
    return out


def binarize(x: torch.Tensor, threshold: float, want_float: bool = True, want_uint8: bool = False):
    """``(x > threshold)`` as x.dtype {0,1} and/or uint8 {0,255} (reference trainer.py:1870-1871, 329-332)."""
    _need_cuda(x)
    if x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError("binarize: fp32 or fp16 input expected, got %s" % x.dtype)
    x = x.contiguous()
    y = _empty_like(x) if want_float else None
    y8 = _empty(x.shape, dtype=torch.uint8, device=x.device) if want_uint8 else None
    lib = _lib.load()
    _lib.check(lib.cgan_binarize(_ptr(x), int(x.dtype == torch.float16), _ptr(y), _ptr(y8), float(threshold),
                                 x.numel(), _stream()), "cgan_binarize")
    if want_float and want_uint8:
        return y, y8
    return y if want_float else y8


def smog(x: torch.Tensor, depth: NHWC, airlight: float, beta: float, alpha: float, yellow_rgb01) -> torch.Tensor:
    """Smog event (reference trainer.py:1879-1939): x NCHW fp32 in [-1, 1], depth the decoder's 1-channel NHWC map;
    returns the smogged image, NCHW fp32."""
    _need_cuda(x, depth.t)
    if depth.c != 1 or depth.cs != 8:
        raise RuntimeError("smog: a one-channel depth map stored with 8 channels is expected")
    x = x.contiguous().float()
    n, c, h, w = x.shape
    if c != 3 or depth.n != n:
        raise RuntimeError("smog: x must be [n,3,h,w] with the depth map's batch size")
    lib = _lib.load()
    nbytes = lib.cgan_smog_workspace_bytes(n)
    ws = _empty(nbytes, dtype=torch.uint8, device=x.device)
    out = _empty_like(x)
    yel = (C.c_float * 3)(*[float(v) for v in yellow_rgb01])
    _lib.check(lib.cgan_smog_nchw(_ptr(x), _ptr(depth.t), depth.dtype_id, _ptr(out), n, h, w, depth.h, depth.w,
                                  float(airlight), float(beta), float(alpha), yel, _ptr(ws), nbytes, _stream()),
               "cgan_smog_nchw")
    return out


def cloudy_cond(x: torch.Tensor, m: torch.Tensor, seg: NHWC, angles: torch.Tensor, sky_idx=9, weight=0.8) -> NHWC:
    """Painter conditioning of paint_cloudy (reference generator.py:319-326): the sky of x replaced by Perlin clouds,
    times (1 - m); ``angles`` [(ry+1), (rx+1)] fp32 device tensor of lattice gradient angles."""
    _need_cuda(x, m, seg.t, angles)
    x = x.contiguous().float()
    m = m.contiguous().float()
    angles = angles.contiguous().float()
    n, _, h, w = x.shape
    ry, rx = angles.shape[0] - 1, angles.shape[1] - 1
    lib = _lib.load()
    nbytes = lib.cgan_cloudy_cond_workspace_bytes(h, w)
    ws = _empty(nbytes, dtype=torch.uint8, device=x.device)
    cond = _empty((n, h, w, 4), dtype=seg.t.dtype, device=x.device)
    _lib.check(lib.cgan_cloudy_cond_nhwc(_ptr(x), _ptr(m), _ptr(seg.t), _ptr(angles), _ptr(cond), seg.dtype_id, n, h, w,
                                         seg.h, seg.w, seg.c, int(sky_idx), ry, rx, float(weight), _ptr(ws), nbytes,
                                         _stream()), "cgan_cloudy_cond_nhwc")
    return NHWC(cond, 3)


def batchnorm_train_stats(x: NHWC, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum):
    """Training-mode nn.BatchNorm2d statistics of ``x`` viewed as ``x.n`` GROUPS of ``x.h * x.w`` pixels each (1 group =
    the whole batch as ONE image of n*h*w pixels; G groups = G equal slices of the batch normalised independently, as if
    each had gone through the layer in its own forward call): returns fp32 [G, Cs] tensors (batch_mean, batch_rstd, mean',
    rstd') with ``bn(x) == (x - mean') * rstd'`` (gamma / beta folded in; identical to the batch pair without them) and
    updates the running statistics (momentum, unbiased variance; once per group, in order) and the step counter (+G) in
    the same two launches."""
    _need_cuda(x.t)
    lib = _lib.load()
    cs = x.t.shape[-1]
    d = NormStatsDesc(x.dtype_id, x.n, x.h * x.w, x.c, float(eps))
    ws_bytes = lib.cgan_instnorm_stats_workspace_bytes(C.byref(d))
    ws = _empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.t.device)
    stats = _empty((4, x.n, cs), dtype=torch.float32, device=x.t.device)
    _lib.check(lib.cgan_batchnorm_train_stats(
        _ptr(x.t), _ptr(gamma), _ptr(beta), float(momentum), _ptr(running_mean), _ptr(running_var),
        _ptr(num_batches_tracked), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]), C.byref(d), _ptr(ws),
        ws_bytes, _stream()), "cgan_batchnorm_train_stats")
    touch(running_mean, running_var, num_batches_tracked)
    return stats[0], stats[1], stats[2], stats[3]


def bn_eval_stats(bn, n: int):
    """(mean, rstd) fp32 [n, Cs] such that eval-mode ``bn(x) == (x - mean) * rstd`` (affine folded in)."""
    rm, rv = bn.running_mean, bn.running_var
    _need_cuda(rm, rv)
    c = rm.numel()
    g = bn.weight.detach().float().contiguous() if getattr(bn, "affine", False) and bn.weight is not None else None
    b = bn.bias.detach().float().contiguous() if getattr(bn, "affine", False) and bn.bias is not None else None
    mean = _empty((n, cs8(c)), dtype=torch.float32, device=rm.device)
    rstd = _empty_like(mean)
    lib = _lib.load()
    _lib.check(lib.cgan_bn_eval_stats(_ptr(g), _ptr(b), _ptr(rm.float().contiguous()), _ptr(rv.float().contiguous()),
                                      float(bn.eps), _ptr(mean), _ptr(rstd), n, c, _stream()), "cgan_bn_eval_stats")
    return mean, rstd


def make_m_cond_bwd(dcond: NHWC, d: NHWC, s: NHWC, with_x: bool):
    """Gradients of ``make_m_cond`` w.r.t. the depth and segmentation maps (the image channels are data)."""
    _need_cuda(dcond.t, d.t, s.t)
    if dcond.c != 1 + s.c + (3 if with_x else 0) or (dcond.n, dcond.h, dcond.w) != (s.n, s.h, s.w):
        raise RuntimeError("make_m_cond_bwd: the conditioning gradient does not match d / s")
    dd, ds = _empty_like(d.t), _empty_like(s.t)
    lib = _lib.load()
    _lib.check(lib.cgan_make_m_cond_bwd_nhwc(_ptr(dcond.t), _ptr(d.t), _ptr(s.t), _ptr(dd), _ptr(ds), d.dtype_id, d.n, d.h,
                                             d.w, s.c, int(bool(with_x)), _stream()), "cgan_make_m_cond_bwd_nhwc")
    return NHWC(dd, 1), NHWC(ds, s.c)


def make_m_cond(d: NHWC, s: NHWC, x: Optional[torch.Tensor]) -> NHWC:
    """cat[normalize(d), softmax(s), bilinear(x)] (reference generator.py:196-230) as an NHWC conditioning map; split maps in,
    split map out (every term in fp32: cgan_pair_make_m_cond)."""
    _need_cuda(d.t, s.t, x)
    if isinstance(d, PairMap) or isinstance(s, PairMap):
        if not (isinstance(d, PairMap) and isinstance(s, PairMap)) or d.t.dtype != s.t.dtype:
            raise RuntimeError("make_m_cond: depth and segmentation must both be split maps of one type")
        _plain_pair(d, "make_m_cond")
        _plain_pair(s, "make_m_cond")
        if (d.h, d.w, d.n) != (s.h, s.w, s.n) or d.c != 1:
            raise RuntimeError("make_m_cond: d and s must share batch and spatial size, d with one channel")
        cond_c = 1 + s.c + (3 if x is not None else 0)
        xx = x.contiguous().float() if x is not None else None
        lib = _lib.load()
        nbytes = lib.cgan_pair_make_m_cond_workspace_bytes(d.n)
        ws = _empty(nbytes, dtype=torch.uint8, device=d.t.device)
        cond = _empty((d.n, d.h, d.w, d.nb * cs8(cond_c)), dtype=d.t.dtype, device=d.t.device)
        _lib.check(lib.cgan_pair_make_m_cond(_ptr(d.t), _ptr(s.t), _ptr(xx), _ptr(cond), d.dtype_id, d.n, d.h, d.w, s.c,
                                             xx.shape[-2] if xx is not None else 0, xx.shape[-1] if xx is not None else 0,
                                             _ptr(ws), nbytes, _stream()), "cgan_pair_make_m_cond")
        return PairMap(cond, cond_c)
    if (d.h, d.w, d.n) != (s.h, s.w, s.n) or d.c != 1:
        raise RuntimeError("make_m_cond: d and s must share batch and spatial size, d with one channel")
    cond_c = 1 + s.c + (3 if x is not None else 0)
    xx = x.contiguous().float() if x is not None else None
    lib = _lib.load()
    nbytes = lib.cgan_make_m_cond_workspace_bytes(d.n)
    ws = _empty(nbytes, dtype=torch.uint8, device=d.t.device)
    cond = _empty((d.n, d.h, d.w, cs4(cond_c)), dtype=d.t.dtype, device=d.t.device)
    _lib.check(lib.cgan_make_m_cond_nhwc(_ptr(d.t), _ptr(s.t), _ptr(xx), _ptr(cond), d.dtype_id, d.n, d.h, d.w, s.c,
                                         xx.shape[-2] if xx is not None else 0, xx.shape[-1] if xx is not None else 0,
                                         _ptr(ws), nbytes, _stream()), "cgan_make_m_cond_nhwc")
    return NHWC(cond, cond_c)


def wildfire(x: torch.Tensor, seg: NHWC, filter_green: float, kernel_size=281, kernel_sigma=140.5, transparency=200,
             crop_bottom=True, sky_idx=9) -> torch.Tensor:
    """Wildfire event (reference fire.py:68-126): x NCHW fp32 in [-1, 1], seg the segmentation decoder's NHWC logits;
    returns the float image in [0, 255] (NCHW) that infer_all then normalises to uint8."""
    _need_cuda(x, seg.t)
    x = x.contiguous().float()
    n, c, h, w = x.shape
    if c != 3 or seg.n != n:
        raise RuntimeError("wildfire: x must be [n,3,h,w] with the segmentation's batch size")
    lib = _lib.load()
    nbytes = lib.cgan_wildfire_workspace_bytes(n, h, w, seg.h, seg.w, int(kernel_size))
    if nbytes == 0:
        _lib.check(-1, "cgan_wildfire_workspace_bytes")
    ws = _empty(nbytes, dtype=torch.uint8, device=x.device)
    out = _empty_like(x)
    _lib.check(lib.cgan_wildfire_nchw(_ptr(x), _ptr(seg.t), seg.dtype_id, _ptr(out), n, h, w, seg.h, seg.w, seg.c,
                                      int(sky_idx), int(kernel_size), float(kernel_sigma), float(transparency),
                                      int(bool(crop_bottom)), float(filter_green), _ptr(ws), nbytes, _stream()),
               "cgan_wildfire_nchw")
    return out


# ------------------------------------------------------------------------------------------------ conv
@dataclass
class PackedConv:
    w: torch.Tensor      # opaque packed weights (uint8)
    bias: torch.Tensor   # fp32 [round_up(c_out,16)]
    c_in: int
    c_out: int
    kh: int
    kw: int
    has_bias: bool
    dtype: torch.dtype
    pair_c_in: int = 0   # > 0: a split-precision operator (pack_conv_weight(pair=True)) of this many logical input channels


def _conv_desc(dtype_id, n, h_in, w_in, c_in, c_out, kh, kw, stride, pad, dil, pad_mode, in_upsample=False,
               act=ACT_NONE, slope=0.2, has_bias=True, has_res=False, res_ups=False) -> ConvDesc:
    h_out = (h_in + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    w_out = (w_in + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    return ConvDesc(dtype_id, n, h_in, w_in, c_in, c_out, kh, kw, stride, pad, dil, pad_mode, h_out, w_out,
                    int(in_upsample), act, slope, int(has_bias), int(has_res), int(res_ups))


def pack_conv_weight(w: torch.Tensor, bias: Optional[torch.Tensor], dtype: torch.dtype,
                     sigma: Optional[torch.Tensor] = None, pair: bool = False) -> PackedConv:
    """fp32 OIHW (+ optional device scalar sigma: packed = w / sigma) -> MFMA fragment order.  ``pair``: the operator of a
    split-precision conv (PairMap input): (W_hi | W_hi | W_lo) along the 3 * round_up(c_in, 8) input channels."""
    _need_cuda(w, bias, sigma)
    w = w.detach().contiguous().float()
    if pair:
        c_out, c_in, kh, kw = w.shape
        w3 = _empty((c_out, split_blocks(dtype) * cs8(c_in), kh, kw), dtype=torch.float32, device=w.device)
        _lib.check(_lib.load().cgan_pair_expand_weight(_ptr(w), _ptr(sigma), _ptr(w3), _DT[dtype], c_out, c_in, kh, kw,
                                                       _stream()), "cgan_pair_expand_weight")
        pk = pack_conv_weight(w3, bias, dtype)
        pk.pair_c_in = c_in
        return pk
    c_out, c_in, kh, kw = w.shape
    d = _conv_desc(_DT[dtype], 1, max(kh, 1), max(kw, 1), c_in, c_out, kh, kw, 1, 0, 1, PAD_ZERO)
    lib = _lib.load()
    nbytes = lib.cgan_conv2d_packed_weight_bytes(C.byref(d))
    if nbytes == 0:
        _lib.check(-1, "cgan_conv2d_packed_weight_bytes")
    packed = _empty(nbytes, dtype=torch.uint8, device=w.device)
    bias_out = _empty(((c_out + 7) // 8 * 8 + 15) // 16 * 16, dtype=torch.float32, device=w.device)
    b = bias.detach().contiguous().float() if bias is not None else None
    _lib.check(lib.cgan_conv2d_pack_weight(_ptr(w), _ptr(b), _ptr(sigma), _ptr(packed), _ptr(bias_out), C.byref(d),
                                           _stream()), "cgan_conv2d_pack_weight")
    return PackedConv(packed, bias_out, c_in, c_out, kh, kw, bias is not None, dtype)


def pack_conv_weights_batched(params, dtype: torch.dtype, reuse=None):
    """``pack_conv_weight`` for a list of (weight, bias-or-None) fp32 OIHW device tensors in ONE launch
    (cgan_conv2d_pack_weight_batched: same arithmetic, bit for bit).  ``reuse[i]``: a PackedConv of the same shape whose
    buffers are overwritten instead of allocated (nothing may still be reading them on another stream)."""
    if not params:
        return []
    lib = _lib.load()
    dev = params[0][0].device
    items = (PackItem * len(params))()
    out, keep, max_frag = [], [], 0
    for i, (w, bias) in enumerate(params):
        _need_cuda(w, bias)
        w = w.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.contiguous().float()
        b = bias.detach() if bias is not None else None
        if b is not None and (b.dtype != torch.float32 or not b.is_contiguous()):
            b = b.contiguous().float()
        keep.append((w, b))
        c_out, c_in, kh, kw = w.shape
        d = _conv_desc(_DT[dtype], 1, max(kh, 1), max(kw, 1), c_in, c_out, kh, kw, 1, 0, 1, PAD_ZERO)
        nbytes = lib.cgan_conv2d_packed_weight_bytes(C.byref(d))
        if nbytes == 0:
            _lib.check(-1, "cgan_conv2d_packed_weight_bytes")
        pk = reuse[i] if reuse is not None else None
        if pk is None or pk.w.numel() != nbytes or pk.dtype != dtype or (pk.c_out, pk.c_in, pk.kh, pk.kw) != (c_out, c_in, kh, kw) \
                or pk.has_bias != (b is not None) or pk.w.device != dev:
            pk = PackedConv(_empty(nbytes, dtype=torch.uint8, device=dev),
                            _empty(((c_out + 7) // 8 * 8 + 15) // 16 * 16, dtype=torch.float32, device=dev),
                            c_in, c_out, kh, kw, b is not None, dtype)
        out.append(pk)
        max_frag = max(max_frag, nbytes // 16)
        items[i] = PackItem(w.data_ptr(), b.data_ptr() if b is not None else 0, 0, pk.w.data_ptr(), pk.bias.data_ptr(),
                            c_out, c_in, kh, kw)
    host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).pin_memory()
    table = host.to(dev, non_blocking=True)
    if _lib.CALL_LOG is not None:
        _lib.log_bytes(sum(w.numel() * 4 for w, _ in params) + sum(pk.w.numel() for pk in out))
    _lib.check(lib.cgan_conv2d_pack_weight_batched(_ptr(table), len(params), _DT[dtype], max_frag, _stream()),
               "cgan_conv2d_pack_weight_batched")
    _PACK_TABLES.append((host, table, keep))          # alive until the copy / kernel that read them are long done
    del _PACK_TABLES[:-8]
    return out


_PACK_TABLES = []

CONV_WS_BYTES = 64 << 20
CONV_WS_KEEP = 8          # (library, device, stream) bindings kept alive; the least recently bound one is released beyond that
_CONV_WS = {}             # key -> (library, buffer or None when the library refused the binding: split-K off for that stream)


def _conv_ws() -> None:
    """Bind (once per library build, device and stream) the calling stream's split-K scratch buffer for the convolution entry
    points (cgan_conv2d_bind_workspace): the small-grid / long-K layers then run as K slices of the LDS-tiled GEMM.  A refused
    binding is not an error -- those layers run on the general kernel, as include/climategan_hip.h documents.  At most
    CONV_WS_KEEP buffers stay allocated: a process that walks through many streams (every Trainer takes a side stream from
    torch's pool of 32 per device) unbinds and frees the oldest instead of pinning 64 MiB per stream for ever."""
    lib = _lib.load()
    st = _stream()
    dev = _GET_DEVICE() if _GET_DEVICE is not None else torch.cuda.current_device()
    key = (id(lib), st.value, dev)
    if key in _CONV_WS:
        if len(_CONV_WS) >= CONV_WS_KEEP:
            _CONV_WS[key] = _CONV_WS.pop(key)      # most recently used last: the streams in use are never the ones released
        return
    while len(_CONV_WS) >= CONV_WS_KEEP:
        okey, (olib, obuf) = next(iter(_CONV_WS.items()))
        if obuf is not None:
            # unbind in the library that holds the binding (the product and the development build keep separate tables); the
            # buffer goes back to the allocator of the stream it was taken on: reuse is ordered after that stream's launches
            with torch.cuda.device(okey[2]):
                olib.cgan_conv2d_bind_workspace(C.c_void_p(okey[1]), C.c_void_p(0), C.c_size_t(0))
        del _CONV_WS[okey]
    buf = _empty(CONV_WS_BYTES, dtype=torch.uint8, device="cuda")
    rc = lib.cgan_conv2d_bind_workspace(st, C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel()))
    _CONV_WS[key] = (lib, buf if rc == 0 else None)


@_batch_chunked("x", "residual", out_bytes_per_sample=lambda g: (lambda hw: hw[0] * hw[1] * cs8(g("pw").c_out) * 2 * _nbf(g("x")))(
    _conv_out_hw(g("x").h * (2 if g("in_upsample") else 1), g("x").w * (2 if g("in_upsample") else 1), g("pw").kh, g("stride"),
                 g("pad"), g("dilation"))))
def conv2d(x: NHWC, pw: PackedConv, stride=1, pad=0, dilation=1, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.2,
           residual: Optional[NHWC] = None, in_upsample=False, residual_upsample=False) -> NHWC:
    """y = act(conv(x) + bias + residual) on NHWC tensors; x may be read through a folded x2 nearest upsample."""
    if isinstance(x, PairMap):
        _plain_pair(x, "conv2d")
        if pw.pair_c_in != x.c or x.t.dtype != pw.dtype:
            raise RuntimeError("conv2d: a pair map of %d channels needs a pair-packed operator of the same type (got %d)"
                               % (x.c, pw.pair_c_in))
        if residual is not None and (not isinstance(residual, PairMap) or residual.c != pw.c_out):
            raise RuntimeError("conv2d: the residual of a pair conv must be a pair map of the output's channels")
        h_in, w_in = (x.h * 2, x.w * 2) if in_upsample else (x.h, x.w)
        d = _conv_desc(x.dtype_id, x.n, h_in, w_in, x.kb * x.cs, pw.c_out, pw.kh, pw.kw, stride, pad, dilation, pad_mode,
                       in_upsample, act, slope, pw.has_bias, residual is not None, residual_upsample)
        y = _empty((x.n, d.h_out, d.w_out, x.nb * cs8(pw.c_out)), dtype=x.t.dtype, device=x.t.device)
        _lib.check(_lib.load().cgan_conv2d_nhwc_fwd_pair(_ptr(x.t), _ptr(pw.w), _ptr(pw.bias),
                                                         _ptr(residual.t if residual is not None else None), _ptr(y),
                                                         C.byref(d), _stream()), "cgan_conv2d_nhwc_fwd_pair")
        return PairMap(y, pw.c_out)
    if pw.pair_c_in:
        raise RuntimeError("conv2d: a pair-packed operator needs a pair map input")
    _need_cuda(x.t)
    _need_cs8("conv2d", x, residual)
    if x.c != pw.c_in:
        raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (x.c, pw.c_in))
    if x.t.dtype != pw.dtype:
        raise RuntimeError("conv2d: activation dtype %s != packed weight dtype %s" % (x.t.dtype, pw.dtype))
    h_in, w_in = (x.h * 2, x.w * 2) if in_upsample else (x.h, x.w)
    d = _conv_desc(x.dtype_id, x.n, h_in, w_in, pw.c_in, pw.c_out, pw.kh, pw.kw, stride, pad, dilation, pad_mode,
                   in_upsample, act, slope, pw.has_bias, residual is not None, residual_upsample)
    if residual is not None:
        rh, rw = (residual.h * 2, residual.w * 2) if residual_upsample else (residual.h, residual.w)
        if (rh, rw) != (d.h_out, d.w_out) or residual.c != pw.c_out or residual.n != x.n:
            raise RuntimeError("conv2d: residual shape mismatch")
    y = _empty((x.n, d.h_out, d.w_out, cs8(pw.c_out)), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _conv_ws()
    _lib.check(lib.cgan_conv2d_nhwc_fwd(_ptr(x.t), _ptr(pw.w), _ptr(pw.bias), _ptr(residual.t if residual else None),
                                        _ptr(y), C.byref(d), _stream()), "cgan_conv2d_nhwc_fwd")
    return NHWC(y, pw.c_out)


class DgradPack:
    """A stride-1 convolution's data-gradient weights, registered during the forward so that ALL of a backward pass's
    operators can be packed in one launch (``dgrad_prepack_run``) instead of one launch per layer inside the backward
    (274 per joint train step).  ``packed`` stays None until that launch; ``conv2d_bwd_data`` then packs on its own."""
    __slots__ = ("w", "sigma", "dtype", "packed")

    def __init__(self, w, sigma, dtype):
        self.w, self.sigma, self.dtype, self.packed = w, sigma, dtype, None


_DGRAD_PENDING = []


def dgrad_register(w: torch.Tensor, sigma, dtype, stride: int) -> Optional[DgradPack]:
    """Called by the autograd Functions' forward for a conv whose input wants a gradient.  Stride-1 only (strided convs
    use the parity-class pack).  The list is bounded: nobody may ever call ``dgrad_prepack_run`` (plain ``backward()``)."""
    if stride != 1 or w.dtype != torch.float32 or not w.is_contiguous():
        return None
    if len(_DGRAD_PENDING) >= 4096:
        del _DGRAD_PENDING[:2048]
    h = DgradPack(w.detach(), sigma, dtype)
    _DGRAD_PENDING.append(h)
    return h


def dgrad_prepack_run(also_used_on=None) -> int:
    """Pack every registered, not yet packed operator (one launch per 16-bit type) and forget the list.  ``also_used_on``:
    a second stream whose kernels will read the packed buffers (the trainer's side stream): the caching allocator is told,
    so that a buffer released by the backward is not handed out again while that stream still reads it."""
    todo = [h for h in _DGRAD_PENDING if h.packed is None]
    del _DGRAD_PENDING[:]
    lib = _lib.load()
    for dtype in (torch.bfloat16, torch.float16):
        hs = [h for h in todo if h.dtype == dtype]
        if not hs:
            continue
        dev = hs[0].w.device
        items = (PackItem * len(hs))()
        max_frag = 0
        for i, h in enumerate(hs):
            c_out, c_in, kh, kw = h.w.shape
            d = _conv_desc(_DT[dtype], 1, max(kh, 1), max(kw, 1), c_out, c_in, kh, kw, 1, 0, 1, PAD_ZERO)   # the operator's dims
            nbytes = lib.cgan_conv2d_packed_weight_bytes(C.byref(d))
            if nbytes == 0:
                _lib.check(-1, "cgan_conv2d_packed_weight_bytes")
            h.packed = _empty(nbytes, dtype=torch.uint8, device=dev)
            max_frag = max(max_frag, nbytes // 16)
            items[i] = PackItem(h.w.data_ptr(), 0, h.sigma.data_ptr() if h.sigma is not None else 0, h.packed.data_ptr(), 0,
                                c_out, c_in, kh, kw, 1)
        host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).pin_memory()
        table = host.to(dev, non_blocking=True)
        if _lib.CALL_LOG is not None:
            _lib.log_bytes(sum(h.w.numel() * 4 + h.packed.numel() for h in hs))
        _lib.check(lib.cgan_conv2d_pack_weight_batched(_ptr(table), len(hs), _DT[dtype], max_frag, _stream()),
                   "cgan_conv2d_pack_weight_batched")
        if also_used_on is not None:
            for h in hs:
                h.packed.record_stream(also_used_on)
        _PACK_TABLES.append((host, table, [(h.w, h.sigma) for h in hs]))   # the launch's sources, not its outputs
        del _PACK_TABLES[:-8]
    return len(todo)


@dataclass
class ConvStats:
    """Per-chunk (mean, M2) rows of a conv output, written by the conv kernel's epilogue (``conv2d_with_stats``)."""
    partial: torch.Tensor      # fp32 [npix / chunk_pixels, Cs, 2]
    chunk_pixels: int


def conv2d_with_stats(x: NHWC, pw: PackedConv, stride=1, pad=0, dilation=1, pad_mode=PAD_ZERO, groups: int = 1):
    """(y, stats): ``conv2d`` without activation / residual whose kernel ALSO leaves the training-mode BatchNorm
    statistics of y as per-chunk partials (cgan_conv2d_nhwc_fwd_stats); stats is None when this descriptor's kernel has
    no such epilogue, or the batch's ``groups`` slices (autograd.bn_groups) are not whole numbers of chunks -- the caller
    then takes the separate statistics pass."""
    _need_cuda(x.t)
    _need_cs8("conv2d", x)
    if x.c != pw.c_in:
        raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (x.c, pw.c_in))
    if x.t.dtype != pw.dtype:
        raise RuntimeError("conv2d: activation dtype %s != packed weight dtype %s" % (x.t.dtype, pw.dtype))
    d = _conv_desc(x.dtype_id, x.n, x.h, x.w, pw.c_in, pw.c_out, pw.kh, pw.kw, stride, pad, dilation, pad_mode,
                   has_bias=pw.has_bias)
    lib = _lib.load()
    ppb = lib.cgan_conv2d_stats_chunk_pixels(C.byref(d))
    npix = x.n * d.h_out * d.w_out
    if ppb <= 0 or x.n % groups or (npix // groups) % ppb:
        return conv2d(x, pw, stride=stride, pad=pad, dilation=dilation, pad_mode=pad_mode), None
    y = _empty((x.n, d.h_out, d.w_out, cs8(pw.c_out)), dtype=x.t.dtype, device=x.t.device)
    partial = _empty((npix // ppb, cs8(pw.c_out), 2), dtype=torch.float32, device=x.t.device)
    _lib.check(lib.cgan_conv2d_nhwc_fwd_stats(_ptr(x.t), _ptr(pw.w), _ptr(pw.bias), _ptr(y), _ptr(partial),
                                              partial.numel() * 4, C.byref(d), _stream()), "cgan_conv2d_nhwc_fwd_stats")
    return NHWC(y, pw.c_out), ConvStats(partial, ppb)


def batchnorm_train_stats_from_partials(st: ConvStats, groups: int, pix_per_group: int, c: int, gamma, beta, running_mean,
                                        running_var, num_batches_tracked, eps, momentum):
    """``batchnorm_train_stats`` from a conv epilogue's partials: same four [G, Cs] outputs, same running-statistics
    updates and step counter, one launch, no pass over the activations."""
    lib = _lib.load()
    cs = st.partial.shape[1]
    d = NormStatsDesc(CGAN_BF16, groups, pix_per_group, c, float(eps))
    stats = _empty((4, groups, cs), dtype=torch.float32, device=st.partial.device)
    _lib.check(lib.cgan_batchnorm_train_stats_from_partials(
        _ptr(st.partial), int(st.chunk_pixels), _ptr(gamma), _ptr(beta), float(momentum), _ptr(running_mean),
        _ptr(running_var), _ptr(num_batches_tracked), _ptr(stats[0]), _ptr(stats[1]), _ptr(stats[2]), _ptr(stats[3]),
        C.byref(d), _stream()), "cgan_batchnorm_train_stats_from_partials")
    touch(running_mean, running_var, num_batches_tracked)
    return stats[0], stats[1], stats[2], stats[3]


@_batch_chunked("dxp")
def reflect_pad_bwd(dxp: NHWC, pad: int) -> NHWC:
    """Backward of nn.ReflectionPad2d(pad): folds the gradient of the padded tensor onto the unpadded extent."""
    _need_cuda(dxp.t)
    h, w = dxp.h - 2 * pad, dxp.w - 2 * pad
    dx = _empty((dxp.n, h, w, dxp.cs), dtype=dxp.t.dtype, device=dxp.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_reflect_pad_bwd_nhwc(_ptr(dxp.t), _ptr(dx), dxp.dtype_id, dxp.n, dxp.c, h, w, int(pad), _stream()),
               "cgan_reflect_pad_bwd_nhwc")
    return NHWC(dx, dxp.c)


def conv2d_bwd_data(dy: NHWC, w: torch.Tensor, x_shape, stride=1, pad=0, dilation=1,
                    sigma: Optional[torch.Tensor] = None, pad_mode=PAD_ZERO, add: Optional[NHWC] = None,
                    prepacked: Optional[DgradPack] = None, relu_out: Optional[NHWC] = None) -> NHWC:
    """dx of y = conv(x, w / sigma): ``w`` fp32 OIHW, ``x_shape`` = (n, h_in, w_in) of the forward input.  Reflect
    padding: data gradient of the pad-0 conv over the padded extent, folded back by the reflection's adjoint.
    ``add``: another gradient contribution of the same input tensor, summed in the kernel's epilogue (stride-1 'same'
    convolutions with zero padding; otherwise by a separate pass)."""
    if relu_out is not None:
        # ``relu_out``: the forward INPUT map, itself the output of a ReLU -- the kernel's epilogue applies that ReLU's
        # derivative (dx * [relu_out > 0]); where the fused form does not apply, the separate pass does
        # (with ``add``: dx = [relu_out > 0] * (data gradient + add), cgan_conv2d_nhwc_bwd_data_add_relu)
        if not (stride == 1 and pad_mode == PAD_ZERO and 2 * pad == dilation * (w.shape[2] - 1) and w.shape[2] == w.shape[3]):
            dx = conv2d_bwd_data(dy, w, x_shape, stride, pad, dilation, sigma, pad_mode, add=add, prepacked=prepacked)
            return act_bwd(relu_out, dx, ACT_RELU)
    per_sample = max(dy.t.nbytes // max(dy.n, 1), (x_shape[1] + 2 * pad) * (x_shape[2] + 2 * pad) * cs8(w.shape[1]) * 2)
    if x_shape[0] * per_sample >= MAX_MAP_BYTES:                 # see _batch_chunked
        k = (MAX_MAP_BYTES - 1) // per_sample
        if k < 1:
            raise RuntimeError("conv2d_bwd_data: ONE sample's map exceeds the 2 GiB a C-ABI call covers")
        return _run_chunks(x_shape[0], k, lambda lo, cnt: conv2d_bwd_data(
            NHWC(dy.t[lo:lo + cnt], dy.c), w, (cnt, x_shape[1], x_shape[2]), stride, pad, dilation, sigma, pad_mode,
            NHWC(add.t[lo:lo + cnt], add.c) if add is not None else None, prepacked,
            NHWC(relu_out.t[lo:lo + cnt], relu_out.c) if relu_out is not None else None))
    if add is not None and not (stride == 1 and pad_mode == PAD_ZERO and 2 * pad == dilation * (w.shape[2] - 1)
                                and w.shape[2] == w.shape[3]):
        dx = conv2d_bwd_data(dy, w, x_shape, stride, pad, dilation, sigma, pad_mode, prepacked=prepacked)
        return NHWC(dx.t + add.t, dx.c)
    if pad_mode == PAD_REFLECT and pad > 0:
        n, h_in, w_in = x_shape
        dxp = conv2d_bwd_data(dy, w, (n, h_in + 2 * pad, w_in + 2 * pad), stride=stride, pad=0, dilation=dilation,
                              sigma=sigma, prepacked=prepacked)
        return reflect_pad_bwd(dxp, pad)
    _need_cuda(dy.t, w, sigma)
    _need_cs8("conv2d_bwd_data", dy)
    w = w.detach().contiguous().float()
    c_out, c_in, kh, kw = w.shape
    n, h_in, w_in = x_shape
    d = _conv_desc(dy.dtype_id, n, h_in, w_in, c_in, c_out, kh, kw, stride, pad, dilation, PAD_ZERO, has_bias=False)
    if (d.h_out, d.w_out) != (dy.h, dy.w) or dy.c != c_out or dy.n != n:
        raise RuntimeError("conv2d_bwd_data: dy shape %s does not match the forward conv" % (tuple(dy.t.shape),))
    lib = _lib.load()
    nbytes = lib.cgan_conv2d_dgrad_packed_weight_bytes(C.byref(d))
    if nbytes == 0:
        _lib.check(-1, "cgan_conv2d_dgrad_packed_weight_bytes")
    if (prepacked is not None and prepacked.packed is not None and stride == 1 and prepacked.dtype == dy.t.dtype
            and prepacked.packed.numel() == nbytes and tuple(prepacked.w.shape) == tuple(w.shape)):
        packed = prepacked.packed                            # packed with the whole backward's operators in one launch
    else:
        packed = _empty(nbytes, dtype=torch.uint8, device=w.device)
        _lib.check(lib.cgan_conv2d_pack_weight_dgrad(_ptr(w), _ptr(sigma), _ptr(packed), C.byref(d), _stream()),
                   "cgan_conv2d_pack_weight_dgrad")
    dx = _empty((n, h_in, w_in, cs8(c_in)), dtype=dy.t.dtype, device=dy.t.device)
    _conv_ws()
    if relu_out is not None:
        if relu_out.t.shape != dx.shape or relu_out.t.dtype != dx.dtype or not relu_out.t.is_contiguous():
            raise RuntimeError("conv2d_bwd_data: ``relu_out`` %s does not match dx %s" % (tuple(relu_out.t.shape), tuple(dx.shape)))
        if add is not None:
            if add.t.shape != dx.shape or add.t.dtype != dx.dtype or not add.t.is_contiguous():
                raise RuntimeError("conv2d_bwd_data: ``add`` %s does not match dx %s" % (tuple(add.t.shape), tuple(dx.shape)))
            _lib.check(lib.cgan_conv2d_nhwc_bwd_data_add_relu(_ptr(dy.t), _ptr(packed), _ptr(add.t), _ptr(relu_out.t), _ptr(dx),
                                                              C.byref(d), _stream()), "cgan_conv2d_nhwc_bwd_data_add_relu")
            return NHWC(dx, c_in)
        _lib.check(lib.cgan_conv2d_nhwc_bwd_data_relu(_ptr(dy.t), _ptr(packed), _ptr(relu_out.t), _ptr(dx), C.byref(d), _stream()),
                   "cgan_conv2d_nhwc_bwd_data_relu")
        return NHWC(dx, c_in)
    if add is not None:
        if add.t.shape != dx.shape or add.t.dtype != dx.dtype or not add.t.is_contiguous():
            raise RuntimeError("conv2d_bwd_data: ``add`` %s does not match dx %s" % (tuple(add.t.shape), tuple(dx.shape)))
        _lib.check(lib.cgan_conv2d_nhwc_bwd_data_add(_ptr(dy.t), _ptr(packed), _ptr(add.t), _ptr(dx), C.byref(d), _stream()),
                   "cgan_conv2d_nhwc_bwd_data_add")
        return NHWC(dx, c_in)
    _lib.check(lib.cgan_conv2d_nhwc_bwd_data(_ptr(dy.t), _ptr(packed), _ptr(dx), C.byref(d), _stream()),
               "cgan_conv2d_nhwc_bwd_data")
    return NHWC(dx, c_in)


@_batch_chunked("x")
def sumpool2x2(x: NHWC) -> NHWC:
    """Backward of the nearest x2 upsample: sums of 2x2 blocks."""
    _need_cuda(x.t)
    if x.h % 2 or x.w % 2 or x.cs != cs8(x.c):
        raise RuntimeError("sumpool2x2: even extent and round_up(c, 8) storage expected")
    y = _empty((x.n, x.h // 2, x.w // 2, x.cs), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_sumpool2x2_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h // 2, x.w // 2, _stream()),
               "cgan_sumpool2x2_nhwc")
    return NHWC(y, x.c)


class ZeroArena:
    """One zero-filled fp32 buffer per optimizer update that the weight-gradient calls carve their outputs from (the
    kernels accumulate into zero-initialised dW / db): one fill instead of one per layer (254 per joint step).  The
    gradients handed to autograd are views of it; a fresh arena is made for every update."""

    def __init__(self, numel: int, device):
        self.buf = torch.zeros(int(numel), dtype=torch.float32, device=device)
        self.off = 0

    def take(self, n: int):
        n_al = (n + 63) // 64 * 64                        # 256-byte aligned slices
        if self.off + n_al > self.buf.numel():
            return None
        t = self.buf[self.off:self.off + n]
        self.off += n_al
        return t


_ZERO_ARENA = None


def set_zero_arena(arena):
    """Install (or, with None, remove) the arena ``zeros_f32`` serves from; returns the previous one."""
    global _ZERO_ARENA
    prev, _ZERO_ARENA = _ZERO_ARENA, arena
    return prev


def zeros_f32(n: int, device) -> torch.Tensor:
    if _ZERO_ARENA is not None and _ZERO_ARENA.buf.device == device:
        t = _ZERO_ARENA.take(n)
        if t is not None:
            return t
    return torch.zeros(n, dtype=torch.float32, device=device)


def conv2d_bwd_weight(x: NHWC, dy: NHWC, w_shape, stride=1, pad=0, dilation=1, want_bias=True,
                      dw: Optional[torch.Tensor] = None, dbias: Optional[torch.Tensor] = None, in_upsample=False,
                      pad_mode=PAD_ZERO, use_workspace=True):
    """(dw fp32 OIHW, dbias fp32 [c_out]) of y = conv(x, w) + b; accumulates into ``dw`` / ``dbias`` when given.
    ``in_upsample``: x is the stored (half-resolution) tensor the forward read through the folded x2 upsample."""
    per_sample = max(x.t.nbytes, dy.t.nbytes) // max(x.n, 1)
    if x.n * per_sample >= MAX_MAP_BYTES:                        # see _batch_chunked: the gradient sums over batch slices
        k = (MAX_MAP_BYTES - 1) // per_sample
        if k < 1:
            raise RuntimeError("conv2d_bwd_weight: ONE sample's map exceeds the 2 GiB a C-ABI call covers")
        for lo in range(0, x.n, k):
            dw, dbias = conv2d_bwd_weight(NHWC(x.t[lo:lo + k], x.c), NHWC(dy.t[lo:lo + k], dy.c), w_shape, stride, pad, dilation,
                                          want_bias, dw, dbias, in_upsample, pad_mode, use_workspace)
        return dw, dbias
    _need_cuda(x.t, dy.t, dw, dbias)
    _need_cs8("conv2d_bwd_weight", x, dy)
    c_out, c_in, kh, kw = w_shape
    h_in, w_in = (x.h * 2, x.w * 2) if in_upsample else (x.h, x.w)
    d = _conv_desc(x.dtype_id, x.n, h_in, w_in, c_in, c_out, kh, kw, stride, pad, dilation, pad_mode,
                   in_upsample=in_upsample)
    if (d.h_out, d.w_out) != (dy.h, dy.w) or dy.c != c_out or x.c != c_in or dy.n != x.n:
        raise RuntimeError("conv2d_bwd_weight: shapes do not match the forward conv")
    if dw is None and want_bias and dbias is None:
        nw = c_out * c_in * kh * kw                      # one zero fill for both gradients
        flat = zeros_f32(nw + c_out, x.t.device)
        dw, dbias = flat[:nw].view(c_out, c_in, kh, kw), flat[nw:]
    if dw is None:
        dw = zeros_f32(c_out * c_in * kh * kw, x.t.device).view(c_out, c_in, kh, kw)
    if want_bias and dbias is None:
        dbias = zeros_f32(c_out, x.t.device)
    lib = _lib.load()
    ws_bytes = lib.cgan_conv2d_bwd_weight_workspace_bytes(C.byref(d)) if use_workspace else 0
    ws = _empty(ws_bytes, dtype=torch.uint8, device=x.t.device) if ws_bytes else None
    _lib.check(lib.cgan_conv2d_nhwc_bwd_weight(_ptr(x.t), _ptr(dy.t), _ptr(dw), _ptr(dbias if want_bias else None),
                                               C.byref(d), _ptr(ws), ws_bytes, _stream()),
               "cgan_conv2d_nhwc_bwd_weight")
    return dw, (dbias if want_bias else None)


# ------------------------------------------------------------------------------------------------ norms
@_batch_chunked("x")
def instnorm_stats(x: NHWC, eps: float = 1e-5):
    """Per-(n,c) mean and 1/sqrt(var+eps) (biased var over H*W) -> two fp32 [N, Cs] tensors."""
    if isinstance(x, PairMap):          # split-precision Painter: fp64 two-pass statistics of the summed components
        _plain_pair(x, "instnorm_stats")
        _need_cuda(x.t)
        mean = _empty((x.n, x.cs), dtype=torch.float32, device=x.t.device)
        rstd = _empty((x.n, x.cs), dtype=torch.float32, device=x.t.device)
        _lib.check(_lib.load().cgan_pair_instnorm_stats(_ptr(x.t), _ptr(mean), _ptr(rstd), x.dtype_id, x.n, x.c, x.h * x.w,
                                                        float(eps), _stream()), "cgan_pair_instnorm_stats")
        return mean, rstd
    _need_cuda(x.t)
    d = NormStatsDesc(x.dtype_id, x.n, x.h * x.w, x.c, eps)
    lib = _lib.load()
    ws_bytes = lib.cgan_instnorm_stats_workspace_bytes(C.byref(d))
    ws = _empty(max(ws_bytes, 16), dtype=torch.uint8, device=x.t.device)
    mean = _empty((x.n, x.cs), dtype=torch.float32, device=x.t.device)
    rstd = _empty((x.n, x.cs), dtype=torch.float32, device=x.t.device)
    _lib.check(lib.cgan_instnorm_stats(_ptr(x.t), _ptr(mean), _ptr(rstd), C.byref(d), _ptr(ws), ws_bytes, _stream()),
               "cgan_instnorm_stats")
    return mean, rstd


@_batch_chunked("x", "mean", "rstd", "residual")
def norm_act_apply(x: NHWC, mean, rstd, act=ACT_NONE, slope=0.2, residual: NHWC = None) -> NHWC:
    """y = act((x - mean) * rstd [+ residual])."""
    if isinstance(x, PairMap):          # split-precision inference: fp32 on the sums of the components (cgan_pair_spade_apply)
        if residual is not None:
            raise RuntimeError("norm_act_apply: no residual on a split map")
        _plain_pair(x, "norm_act_apply")
        _need_cuda(mean, rstd)
        y = _empty_like(x.t)
        _lib.check(_lib.load().cgan_pair_spade_apply(_ptr(x.t), _ptr(mean), _ptr(rstd), None, None, _ptr(y), x.dtype_id, x.n, x.h,
                                                     x.w, x.c, 0, int(act), float(slope), _stream()), "cgan_pair_spade_apply")
        return PairMap(y, x.c)
    _need_cuda(x.t, mean, rstd)
    d = NormStatsDesc(x.dtype_id, x.n, x.h * x.w, x.c, 0.0)
    y = _empty_like(x.t)
    lib = _lib.load()
    if residual is not None:
        _need_cuda(residual.t)
        if residual.t.shape != x.t.shape or residual.t.dtype != x.t.dtype or not residual.t.is_contiguous():
            raise ValueError("norm_act_apply: the residual must match x (shape %s, dtype %s, contiguous)"
                             % (tuple(x.t.shape), x.t.dtype))
    _lib.check(lib.cgan_norm_add_act_apply(_ptr(x.t), _ptr(mean), _ptr(rstd),
                                           _ptr(residual.t) if residual is not None else None, _ptr(y), C.byref(d),
                                           act, slope, _stream()), "cgan_norm_add_act_apply")
    return NHWC(y, x.c)


# ------------------------------------------------------------------------------------------------ training path
@_batch_chunked("out", "dy")
def act_bwd(out: NHWC, dy: NHWC, act, slope=0.2) -> NHWC:
    """dx = dy * act'(.) with the derivative taken from the activation's output."""
    _need_cuda(out.t, dy.t)
    dx = _empty_like(out.t)
    lib = _lib.load()
    _lib.check(lib.cgan_act_bwd(_ptr(out.t), _ptr(dy.t), _ptr(dx), out.dtype_id, act, slope, out.t.numel(), _stream()),
               "cgan_act_bwd")
    return NHWC(dx, out.c)


@_batch_chunked("out", "dy", "rstd")
def instnorm_act_bwd(out: NHWC, dy: NHWC, rstd: torch.Tensor, act=ACT_NONE, slope=0.2) -> NHWC:
    """Backward of out = act(instance_norm(x)) given out, dy and the forward's rstd (act none or LeakyReLU)."""
    _need_cuda(out.t, dy.t, rstd)
    d = NormStatsDesc(out.dtype_id, out.n, out.h * out.w, out.c, 0.0)
    lib = _lib.load()
    nbytes = lib.cgan_instnorm_act_bwd_workspace_bytes(C.byref(d))
    ws = _empty(nbytes, dtype=torch.uint8, device=out.t.device)
    dx = _empty_like(out.t)
    _lib.check(lib.cgan_instnorm_act_bwd(_ptr(out.t), _ptr(dy.t), _ptr(rstd), _ptr(dx), C.byref(d), act, slope,
                                         _ptr(ws), nbytes, _stream()), "cgan_instnorm_act_bwd")
    return NHWC(dx, out.c)


@_batch_chunked("x")
def bce_logits(x: NHWC, target: float, weight: float, loss_accum: torch.Tensor, want_grad=True):
    """loss_accum += weight * sum BCEWithLogits(x, target) over x's logical channels; returns d(loss)/dx or None."""
    _need_cuda(x.t, loss_accum)
    dx = _empty_like(x.t) if want_grad else None
    lib = _lib.load()
    _lib.check(lib.cgan_bce_logits_nhwc(_ptr(x.t), x.dtype_id, x.n * x.h * x.w, x.c, float(target), float(weight),
                                        _ptr(loss_accum), _ptr(dx), _stream()), "cgan_bce_logits_nhwc")
    return NHWC(dx, x.c) if want_grad else None


@_batch_chunked("x")
def hinge_loss(x: NHWC, target_is_real: bool, for_discriminator: bool, weight: float, loss_accum: torch.Tensor,
               want_grad=True):
    """loss_accum += weight * sum HingeLoss.loss terms of x's logical channels (reference losses.py:565-579); returns
    d(loss)/dx or None."""
    _need_cuda(x.t, loss_accum)
    dx = _empty_like(x.t) if want_grad else None
    lib = _lib.load()
    _lib.check(lib.cgan_hinge_nhwc(_ptr(x.t), x.dtype_id, x.n * x.h * x.w, x.c, int(bool(target_is_real)),
                                   int(bool(for_discriminator)), float(weight), _ptr(loss_accum), _ptr(dx), _stream()),
               "cgan_hinge_nhwc")
    return NHWC(dx, x.c) if want_grad else None


@_batch_chunked("a", "b")
def l1_loss(a: NHWC, b: NHWC, weight: float, loss_accum: torch.Tensor, want_grad=True):
    """loss_accum += weight * sum|a - b|; returns d(loss)/da or None."""
    _need_cuda(a.t, b.t, loss_accum)
    if a.t.shape != b.t.shape:
        raise RuntimeError("l1_loss: shape mismatch")
    da = _empty_like(a.t) if want_grad else None
    lib = _lib.load()
    _lib.check(lib.cgan_l1_nhwc(_ptr(a.t), _ptr(b.t), a.dtype_id, a.t.numel(), float(weight), _ptr(loss_accum),
                                _ptr(da), _stream()), "cgan_l1_nhwc")
    return NHWC(da, a.c) if want_grad else None


def spectral_norm_bwd(grad_w: torch.Tensor, w_bar: torch.Tensor, u: torch.Tensor, v: torch.Tensor,
                      sigma: torch.Tensor) -> torch.Tensor:
    """In place: gradient w.r.t. w = w_bar / sigma  ->  gradient w.r.t. w_bar (u, v constants)."""
    _need_cuda(grad_w, w_bar, u, v, sigma)
    rows = w_bar.shape[0]
    cols = w_bar.numel() // rows
    scratch = _empty(1024, dtype=torch.float32, device=grad_w.device)      # CGAN_SN_BWD_WORKSPACE_FLOATS
    lib = _lib.load()
    _lib.check(lib.cgan_spectral_norm_bwd(_ptr(grad_w), _ptr(w_bar), _ptr(u), _ptr(v), _ptr(sigma), rows, cols,
                                          _ptr(scratch), _stream()), "cgan_spectral_norm_bwd")
    return grad_w


@dataclass
class PackedSpade:
    buf: torch.Tensor
    c: int
    cond_c: int
    dtype: torch.dtype


def _spade_desc(dtype_id, n, h, w, c, x_ups, cond_h, cond_w, cond_c, act, slope=0.2):
    return SpadeDesc(dtype_id, n, h, w, c, int(x_ups), cond_h, cond_w, cond_c, 128, 3, act, slope)


def pack_spade_weights(w_shared, b_shared, w_gamma, b_gamma, w_beta, b_beta, dtype: torch.dtype) -> PackedSpade:
    _need_cuda(w_shared, w_gamma, w_beta)
    c, hidden = w_gamma.shape[0], w_gamma.shape[1]
    cond_c = w_shared.shape[1]
    if hidden != 128 or w_shared.shape[0] != 128 or tuple(w_gamma.shape[2:]) != (3, 3):
        raise ValueError("SPADE: only hidden=128, kernel_size=3 is supported")
    d = _spade_desc(_DT[dtype], 1, 16, 16, c, False, 16, 16, cond_c, ACT_NONE)
    lib = _lib.load()
    nbytes = lib.cgan_spade_packed_weight_bytes(C.byref(d))
    if nbytes == 0:
        _lib.check(-1, "cgan_spade_packed_weight_bytes")
    buf = _empty(nbytes, dtype=torch.uint8, device=w_gamma.device)
    ts = [t.detach().contiguous().float() for t in (w_shared, b_shared, w_gamma, b_gamma, w_beta, b_beta)]
    _lib.check(lib.cgan_spade_pack_weights(*[_ptr(t) for t in ts], _ptr(buf), C.byref(d), _stream()),
               "cgan_spade_pack_weights")
    return PackedSpade(buf, c, cond_c, dtype)


@_batch_chunked("x", "mean", "rstd", "cond", out_bytes_per_sample=lambda g: g("x").h * g("x").w * g("x").cs * 2 * (4 if g("x_upsample") else 1))
def spade_fused(x: NHWC, mean, rstd, cond: NHWC, pk: PackedSpade, act=ACT_NONE, slope=0.2, x_upsample=False,
                want_gamma=False):
    """Fused SPADE: act((x-mean)*rstd*(1+gamma(cond))+beta(cond)); x optionally read through x2 nearest.
    ``want_gamma`` (training): returns (y, gamma) -- the kernel also writes the modulation map the backward needs."""
    _need_cuda(x.t, cond.t, mean, rstd)
    if x.c != pk.c or cond.c != pk.cond_c:
        raise RuntimeError("spade_fused: channel mismatch (x %d vs %d, cond %d vs %d)" % (x.c, pk.c, cond.c, pk.cond_c))
    if cond.cs != cs4(cond.c):
        raise RuntimeError("spade_fused: cond must be stored with round_up(cond_c,4) channels")
    if x.t.dtype != pk.dtype or cond.t.dtype != pk.dtype:
        raise RuntimeError("spade_fused: dtype mismatch")
    h, w = (x.h * 2, x.w * 2) if x_upsample else (x.h, x.w)
    d = _spade_desc(x.dtype_id, x.n, h, w, x.c, x_upsample, cond.h, cond.w, cond.c, act, slope)
    y = _empty((x.n, h, w, x.cs), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    if want_gamma:
        gamma = _empty_like(y)
        _lib.check(lib.cgan_spade_fused_fwd_train(_ptr(x.t), _ptr(mean), _ptr(rstd), _ptr(cond.t), _ptr(pk.buf), _ptr(y),
                                                  _ptr(gamma), C.byref(d), _stream()), "cgan_spade_fused_fwd_train")
        return NHWC(y, x.c), NHWC(gamma, x.c)
    _lib.check(lib.cgan_spade_fused_fwd(_ptr(x.t), _ptr(mean), _ptr(rstd), _ptr(cond.t), _ptr(pk.buf), _ptr(y),
                                        C.byref(d), _stream()), "cgan_spade_fused_fwd")
    return NHWC(y, x.c)


@_batch_chunked("dy", "y", "x", "mean", "rstd", "gamma", out_bytes_per_sample=lambda g: g("y").h * g("y").w * cs8(2 * g("y").c) * 2)
def spade_bwd_prepare(dy: NHWC, y: NHWC, x: NHWC, mean, rstd, gamma: NHWC, act=ACT_NONE, slope=0.2, x_upsample=False):
    """Elementwise stage of the SPADE backward: returns (dgb [2C channels: d_gamma | d_beta], xhat, dxhat)."""
    _need_cuda(dy.t, y.t, x.t, mean, rstd, gamma.t)
    c = y.c
    d = _spade_desc(y.dtype_id, y.n, y.h, y.w, c, x_upsample, y.h, y.w, 3, act, slope)
    dgb = _empty((y.n, y.h, y.w, cs8(2 * c)), dtype=y.t.dtype, device=y.t.device)
    xhat = _empty_like(y.t)
    dxhat = _empty_like(y.t)
    lib = _lib.load()
    _lib.check(lib.cgan_spade_bwd_prepare(_ptr(dy.t), _ptr(y.t), _ptr(x.t), _ptr(mean), _ptr(rstd), _ptr(gamma.t),
                                          _ptr(dgb), _ptr(xhat), _ptr(dxhat), C.byref(d), _stream()),
               "cgan_spade_bwd_prepare")
    return NHWC(dgb, 2 * c), NHWC(xhat, c), NHWC(dxhat, c)


@_batch_chunked("x", "mean", "rstd", "gamma", "beta")
def pair_spade_apply(x: "PairMap", mean, rstd, gamma: "PairMap", beta: "PairMap", act=ACT_NONE, slope=0.2,
                     x_upsample=False) -> "PairMap":
    """SPADE's de-normalisation on split maps (cgan_pair_spade_apply): y = act((x - mean) rstd (1 + gamma) + beta) in fp32."""
    _need_cuda(x.t, mean, rstd, gamma.t, beta.t)
    h, w = (x.h * 2, x.w * 2) if x_upsample else (x.h, x.w)
    if (gamma.n, gamma.h, gamma.w, gamma.c) != (x.n, h, w, x.c) or gamma.t.shape != beta.t.shape or gamma.t.dtype != x.t.dtype:
        raise RuntimeError("pair_spade_apply: gamma / beta must be pair maps of x's channels at the output extent")
    y = _empty((x.n, h, w, x.nb * x.cs), dtype=x.t.dtype, device=x.t.device)
    _lib.check(_lib.load().cgan_pair_spade_apply(_ptr(x.t), _ptr(mean), _ptr(rstd), _ptr(gamma.t), _ptr(beta.t), _ptr(y), x.dtype_id,
                                                 x.n, h, w, x.c, int(bool(x_upsample)), int(act), float(slope), _stream()),
               "cgan_pair_spade_apply")
    return PairMap(y, x.c)


def spade_hidden_bwd(dgb: NHWC, w_gb: torch.Tensor, seg: NHWC, pw_shared: PackedConv, c: int, want_bias=True):
    """(dw_shared, db_shared) of SPADE's mlp_shared from the gamma||beta gradient map ``dgb`` (spade_bwd_prepare), in ONE
    kernel (cgan_spade_hidden_bwd): the hidden map is re-computed per tile from ``seg`` (the conditioning image at the map's
    extent, <= 4 channels) and its gradient never leaves the chip.  ``w_gb`` = cat[w_gamma, w_beta] fp32 OIHW."""
    _need_cuda(dgb.t, w_gb, seg.t)
    n, h, w = dgb.n, dgb.h, dgb.w
    hidden = w_gb.shape[1]
    if seg.c > 4 or hidden != 128 or (seg.n, seg.h, seg.w) != (n, h, w) or dgb.c != 2 * c or w_gb.shape[0] != 2 * c:
        raise RuntimeError("spade_hidden_bwd: needs a <= 4-channel conditioning image at the map's extent and hidden width 128")
    lib = _lib.load()
    w_gb = w_gb.detach().contiguous().float()
    dconv = _conv_desc(dgb.dtype_id, n, h, w, hidden, 2 * c, 3, 3, 1, 1, 1, PAD_ZERO, has_bias=False)
    nbytes = lib.cgan_conv2d_dgrad_packed_weight_bytes(C.byref(dconv))
    if nbytes == 0:
        _lib.check(-1, "cgan_conv2d_dgrad_packed_weight_bytes")
    packed = _empty(nbytes, dtype=torch.uint8, device=w_gb.device)
    _lib.check(lib.cgan_conv2d_pack_weight_dgrad(_ptr(w_gb), _ptr(None), _ptr(packed), C.byref(dconv), _stream()),
               "cgan_conv2d_pack_weight_dgrad")
    d = _spade_desc(dgb.dtype_id, n, h, w, c, False, h, w, seg.c, ACT_NONE)
    ws = _empty(lib.cgan_spade_hidden_bwd_workspace_bytes(C.byref(d)), dtype=torch.uint8, device=dgb.t.device)
    dw = zeros_f32(hidden * seg.c * 9, dgb.t.device).view(hidden, seg.c, 3, 3)
    db = zeros_f32(hidden, dgb.t.device) if want_bias else None
    _lib.check(lib.cgan_spade_hidden_bwd(_ptr(dgb.t), _ptr(packed), _ptr(seg.t), _ptr(pw_shared.w), _ptr(pw_shared.bias), _ptr(dw),
                                         _ptr(db), _ptr(ws), ws.numel(), C.byref(d), _stream()), "cgan_spade_hidden_bwd")
    return dw, db


@_batch_chunked("fake", "x", "m")
def painter_heads(fake: Optional[NHWC], x: torch.Tensor, m: torch.Tensor, dtype, want_d=True, want_vgg=False):
    """(d_in, vgg_in) of the pasted image p = x (1 - m) + fake m (or p = x when ``fake`` is None): the discriminator
    input [m | p] (4 channels) and vgg_preprocess(p * m), NHWC 16-bit; x, m NCHW fp32.  The VGG input is stored as a
    16-bit pair per colour, channels [b_hi, g_hi, r_hi, b_lo, g_lo, r_lo] with value = hi + lo: its magnitudes (100-150)
    would otherwise lose +-0.5 (bf16) in the store; ``losses.Vgg19`` runs its first conv on the six channels."""
    _need_cuda(x, m, fake.t if fake is not None else None)
    x = x.contiguous().float()
    m = m.contiguous().float()
    n, _, h, w = x.shape
    d_in = _empty((n, h, w, 8), dtype=dtype, device=x.device) if want_d else None
    v_in = _empty((n, h, w, 8), dtype=dtype, device=x.device) if want_vgg else None
    lib = _lib.load()
    _lib.check(lib.cgan_painter_heads_fwd(_ptr(fake.t if fake is not None else None), _ptr(x), _ptr(m), _ptr(d_in),
                                          _ptr(v_in), _DT[dtype], n, h, w, _stream()), "cgan_painter_heads_fwd")
    return (NHWC(d_in, 4) if want_d else None), (NHWC(v_in, 6) if want_vgg else None)


@_batch_chunked("d_d_in", "d_vgg_in", "m")
def painter_heads_bwd(d_d_in: Optional[NHWC], d_vgg_in: Optional[NHWC], m: torch.Tensor) -> NHWC:
    ref = d_d_in if d_d_in is not None else d_vgg_in
    _need_cuda(ref.t, m)
    m = m.contiguous().float()
    n, h, w = ref.n, ref.h, ref.w
    dfake = _empty((n, h, w, 8), dtype=ref.t.dtype, device=ref.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_painter_heads_bwd(_ptr(d_d_in.t if d_d_in is not None else None),
                                          _ptr(d_vgg_in.t if d_vgg_in is not None else None), _ptr(m), _ptr(dfake),
                                          ref.dtype_id, n, h, w, _stream()), "cgan_painter_heads_bwd")
    return NHWC(dfake, 3)


@_batch_chunked("dy", out_bytes_per_sample=lambda g: g("in_hw")[0] * g("in_hw")[1] * g("dy").cs * 2)
def avgpool3x3s2_bwd(dy: NHWC, in_hw) -> NHWC:
    _need_cuda(dy.t)
    h, w = in_hw
    dx = _empty((dy.n, h, w, dy.cs), dtype=dy.t.dtype, device=dy.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_avgpool3x3s2_bwd_nhwc(_ptr(dy.t), _ptr(dx), dy.dtype_id, dy.n, dy.c, h, w, _stream()),
               "cgan_avgpool3x3s2_bwd_nhwc")
    return NHWC(dx, dy.c)


@_batch_chunked("x")
def maxpool2x2(x: NHWC) -> NHWC:
    _need_cuda(x.t)
    y = _empty((x.n, x.h // 2, x.w // 2, x.cs), dtype=x.t.dtype, device=x.t.device)
    lib = _lib.load()
    _lib.check(lib.cgan_maxpool2x2_nhwc(_ptr(x.t), _ptr(y), x.dtype_id, x.n, x.c, x.h, x.w, _stream()),
               "cgan_maxpool2x2_nhwc")
    return NHWC(y, x.c)


@_batch_chunked("x", "dy")
def maxpool2x2_bwd(x: NHWC, dy: NHWC, relu_input=False) -> NHWC:
    """``relu_input``: x is a ReLU's output whose derivative is taken here too (dx * [x > 0], cgan_maxpool2x2_relu_bwd_nhwc)."""
    _need_cuda(x.t, dy.t)
    dx = _empty_like(x.t)
    lib = _lib.load()
    if relu_input:
        _lib.check(lib.cgan_maxpool2x2_relu_bwd_nhwc(_ptr(x.t), _ptr(dy.t), _ptr(dx), x.dtype_id, x.n, x.c, x.h, x.w,
                                                     _stream()), "cgan_maxpool2x2_relu_bwd_nhwc")
    else:
        _lib.check(lib.cgan_maxpool2x2_bwd_nhwc(_ptr(x.t), _ptr(dy.t), _ptr(dx), x.dtype_id, x.n, x.c, x.h, x.w, _stream()),
                   "cgan_maxpool2x2_bwd_nhwc")
    return NHWC(dx, x.c)


# ------------------------------------------------------------------------------------------------ spectral norm
def spectral_norm_power_iter(w_bar: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """One power iteration (reference norms.py:100-112); u, v updated IN PLACE; returns device scalar sigma."""
    _need_cuda(w_bar, u, v)
    if w_bar.dtype != torch.float32 or u.dtype != torch.float32 or v.dtype != torch.float32:
        raise RuntimeError("spectral_norm_power_iter: fp32 parameters expected")
    if not (w_bar.is_contiguous() and u.is_contiguous() and v.is_contiguous()):
        raise RuntimeError("spectral_norm_power_iter: contiguous parameters expected")
    rows = w_bar.shape[0]
    cols = w_bar.numel() // rows
    assert u.numel() == rows and v.numel() == cols
    lib = _lib.load()
    ws_bytes = lib.cgan_spectral_norm_workspace_bytes(rows, cols)
    ws = _empty(ws_bytes, dtype=torch.uint8, device=w_bar.device)
    sigma = _empty(1, dtype=torch.float32, device=w_bar.device)
    _lib.check(lib.cgan_spectral_norm_power_iter(_ptr(w_bar), _ptr(u), _ptr(v), _ptr(sigma), rows, cols, _ptr(ws),
                                                 ws_bytes, _stream()), "cgan_spectral_norm_power_iter")
    return sigma


class SpectralNormGroup:
    """All spectral-norm convs of a network, power-iterated and re-packed together: 4 + 1 launches per forward
    instead of 5 per layer.  Semantics are those of the per-layer path (reference norms.py:100-112,141-143: one
    power iteration per wrapped conv per forward, u/v updated in place, conv weight = w_bar / sigma); the
    arithmetic is bit-identical to ``spectral_norm_power_iter`` + ``pack_conv_weight``.

    ``params``: list of (w_bar, u, v, bias-or-None) fp32 device tensors.  Device-side tables, the workspace and
    the packed-weight buffers are allocated once and reused; they are rebuilt if a parameter moves.
    """

    def __init__(self, params, dtype: torch.dtype):
        self.dtype = dtype
        self.key = self._key(params, dtype)
        lib = _lib.load()
        dev = params[0][0].device
        n = len(params)
        self.n = n
        self.sigma = _empty(n, dtype=torch.float32, device=dev)
        ws_sizes = []
        self.packed = []
        self.max_rows = self.max_cols = self.max_frag = 0
        for (w_bar, u, v, bias) in params:
            _need_cuda(w_bar, u, v, bias)
            for t in (w_bar, u, v, bias):
                if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
                    raise RuntimeError("SpectralNormGroup: contiguous fp32 parameters expected")
            rows = w_bar.shape[0]
            cols = w_bar.numel() // rows
            self.max_rows, self.max_cols = max(self.max_rows, rows), max(self.max_cols, cols)
            ws_sizes.append(lib.cgan_spectral_norm_workspace_bytes(rows, cols))
            c_out, c_in, kh, kw = w_bar.shape
            d = _conv_desc(_DT[dtype], 1, kh, kw, c_in, c_out, kh, kw, 1, 0, 1, PAD_ZERO)
            nbytes = lib.cgan_conv2d_packed_weight_bytes(C.byref(d))
            self.max_frag = max(self.max_frag, nbytes // 16)
            pw = _empty(nbytes, dtype=torch.uint8, device=dev)
            bo = _empty(((c_out + 7) // 8 * 8 + 15) // 16 * 16, dtype=torch.float32, device=dev)
            self.packed.append(PackedConv(pw, bo, c_in, c_out, kh, kw, bias is not None, dtype))
        offs = [0]
        for sz in ws_sizes:
            offs.append(offs[-1] + (sz + 255) // 256 * 256)
        self.ws = _empty(offs[-1], dtype=torch.uint8, device=dev)
        sn_items = (SnItem * n)()
        pk_items = (PackItem * n)()
        for i, (w_bar, u, v, bias) in enumerate(params):
            rows = w_bar.shape[0]
            sig = self.sigma.data_ptr() + 4 * i
            sn_items[i] = SnItem(w_bar.data_ptr(), u.data_ptr(), v.data_ptr(), sig, self.ws.data_ptr() + offs[i], rows,
                                 w_bar.numel() // rows)
            c_out, c_in, kh, kw = w_bar.shape
            pk_items[i] = PackItem(w_bar.data_ptr(), bias.data_ptr() if bias is not None else 0, sig,
                                   self.packed[i].w.data_ptr(), self.packed[i].bias.data_ptr(), c_out, c_in, kh, kw)
        self.sn_table = torch.frombuffer(bytearray(bytes(sn_items)), dtype=torch.uint8).to(dev)
        self.pk_table = torch.frombuffer(bytearray(bytes(pk_items)), dtype=torch.uint8).to(dev)

    @staticmethod
    def _key(params, dtype):
        return tuple((t.data_ptr() if t is not None else 0) for ps in params for t in ps) + (dtype,)

    def matches(self, params, dtype):
        return self.key == self._key(params, dtype)

    def step(self):
        """One power iteration of every layer (u, v updated in place) + re-pack of every w_bar / sigma."""
        lib = _lib.load()
        if _lib.CALL_LOG is not None:            # w_bar is read twice per power iteration (W^T u, W v)
            _lib.log_bytes(2 * sum(pk.c_out * pk.c_in * pk.kh * pk.kw * 4 for pk in self.packed))
        _lib.check(lib.cgan_spectral_norm_power_iter_batched(_ptr(self.sn_table), self.n, self.max_rows, self.max_cols,
                                                             _stream()), "cgan_spectral_norm_power_iter_batched")
        if _lib.CALL_LOG is not None:
            _lib.log_bytes(sum(pk.c_out * pk.c_in * pk.kh * pk.kw * 4 + pk.w.numel() for pk in self.packed))
        _lib.check(lib.cgan_conv2d_pack_weight_batched(_ptr(self.pk_table), self.n, _DT[self.dtype], self.max_frag,
                                                       _stream()), "cgan_conv2d_pack_weight_batched")
        return self.packed


# ------------------------------------------------------------------------------------------------ input pipeline
def _gaussian_taps(sigma: float):
    """scipy.ndimage's 1-D Gaussian taps for ``gaussian_filter(truncate=4)``: radius int(4 sigma + 0.5), weights
    exp(-0.5 x^2 / sigma^2) normalised to sum 1 (float64, numpy)."""
    import numpy as np

    radius = int(4.0 * float(sigma) + 0.5)
    if sigma <= 1e-15 or radius == 0:
        return 0, None
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return radius, phi / phi.sum()


def resize_crop_geometry(h: int, w: int, to: int):
    """(rows, cols, top, left) of apply_events.resize_and_crop for an h x w image (apply_events.py:224-238)."""
    lib = _lib.load()
    r, c, t, l = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _lib.check(lib.cgan_resize_crop_geometry(h, w, to, C.byref(r), C.byref(c), C.byref(t), C.byref(l)),
               "cgan_resize_crop_geometry")
    return r.value, c.value, t.value, l.value


def resize_and_crop_u8(img: torch.Tensor, to: int = 640, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """uint8 HWC device image -> fp32 [3, to, to] in [-1, 1] = to_m1_p1(resize_and_crop(img, to))
    (apply_events.py:179-195, 211-241).  ``out``: an existing [c, to, to] fp32 slot (e.g. one image of a batch)."""
    _need_cuda(img, out)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] not in (1, 3, 4):
        raise RuntimeError("resize_and_crop_u8: a uint8 [H, W, C] image is expected, got %s %s" % (img.dtype, tuple(img.shape)))
    img = img.contiguous()
    h, w, c = img.shape
    rows, cols, _, _ = resize_crop_geometry(h, w, to)
    taps = []
    for scale in (h / rows, w / cols):                       # skimage: sigma = max(0, (factor - 1) / 2)
        radius, wts = _gaussian_taps(max(0.0, (scale - 1.0) / 2.0))
        taps.append((radius, None if wts is None else torch.from_numpy(wts).to(img.device)))
    if out is None:
        out = _empty((c, to, to), dtype=torch.float32, device=img.device)
    elif out.shape != (c, to, to) or out.dtype != torch.float32 or not out.is_contiguous():
        raise RuntimeError("resize_and_crop_u8: out must be a contiguous fp32 [%d, %d, %d] tensor" % (c, to, to))
    lib = _lib.load()
    ws_bytes = lib.cgan_resize_crop_u8_workspace_bytes(h, w, c)
    ws = _empty(ws_bytes, dtype=torch.uint8, device=img.device)
    _lib.check(lib.cgan_resize_crop_u8(_ptr(img), h, w, c, to, _ptr(taps[0][1]), taps[0][0], _ptr(taps[1][1]),
                                       taps[1][0], _ptr(out), _ptr(ws), ws_bytes, _stream()), "cgan_resize_crop_u8")
    return out


def resize_u8(img: torch.Tensor, size) -> torch.Tensor:
    """uint8 HWC device image -> fp32 [C, rows, cols] in [-1, 1] = to_m1_p1(resize(img, size, anti_aliasing=True)), the
    keep_ratio branch of apply_events (apply_events.py:494-497, 502): no crop, no uint8 truncation."""
    _need_cuda(img)
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] not in (1, 3, 4):
        raise RuntimeError("resize_u8: a uint8 [H, W, C] image is expected, got %s %s" % (img.dtype, tuple(img.shape)))
    img = img.contiguous()
    h, w, c = img.shape
    rows, cols = int(size[0]), int(size[1])
    if rows <= 0 or cols <= 0:
        raise RuntimeError("resize_u8: empty output size %s" % (size,))
    taps = []
    for scale in (h / rows, w / cols):
        radius, wts = _gaussian_taps(max(0.0, (scale - 1.0) / 2.0))
        taps.append((radius, None if wts is None else torch.from_numpy(wts).to(img.device)))
    out = _empty((c, rows, cols), dtype=torch.float32, device=img.device)
    lib = _lib.load()
    ws_bytes = lib.cgan_resize_crop_u8_workspace_bytes(h, w, c)
    ws = _empty(ws_bytes, dtype=torch.uint8, device=img.device)
    _lib.check(lib.cgan_resize_u8(_ptr(img), h, w, c, rows, cols, _ptr(taps[0][1]), taps[0][0], _ptr(taps[1][1]), taps[1][0],
                                  _ptr(out), _ptr(ws), ws_bytes, _stream()), "cgan_resize_u8")
    return out
