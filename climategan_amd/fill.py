"""Portable deterministic tensor fill (synthetic weights and inputs for bench.py, smoke() and the tests).

A counter-based hash (splitmix64 finaliser) of (seed, element index) -> uniform values.  It does not
depend on torch's RNG, so the dev container (which can import the reference) and the GPU box (which
cannot) regenerate bit-identical weights and inputs; golden fixtures then only need to hold outputs.
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def key_seed(key: str, seed: int = 0) -> int:
    """Stable 32-bit seed from a state-dict key."""
    return (zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF


def uniform01(shape, seed: int) -> np.ndarray:
    """float64 uniform in [0,1) with 32 bits of randomness per element."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        h = _splitmix64(idx ^ (_splitmix64(np.array([seed], dtype=np.uint64)) & _M64))
    u = (h >> np.uint64(32)).astype(np.float64) / 4294967296.0
    return u.reshape(shape)


def uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    return (lo + (hi - lo) * uniform01(shape, seed)).astype(np.float32)


def rect_mask(batch: int, h: int, w: int, seed: int = 0) -> np.ndarray:
    """Binary {0,1} mask [B,1,H,W]: union of 3 axis-aligned rectangles (~35% coverage), per image."""
    m = np.zeros((batch, 1, h, w), dtype=np.float32)
    u = uniform01((batch, 3, 4), seed)
    for b in range(batch):
        for r in range(3):
            y0 = int(u[b, r, 0] * h * 0.6)
            x0 = int(u[b, r, 1] * w * 0.6)
            hh = int((0.2 + 0.25 * u[b, r, 2]) * h)
            ww = int((0.2 + 0.25 * u[b, r, 3]) * w)
            m[b, 0, y0 : y0 + hh, x0 : x0 + ww] = 1.0
    return m


def fill_state_dict(shapes: dict, seed: int = 0, gain: float = 1.0, res_gamma: float = 1.0) -> dict:
    """Fill a {key: shape} dict the way an (untrained) reference module is populated.

    * conv weights / ``weight_bar``: U(-1/sqrt(fan_in), 1/sqrt(fan_in))  (torch's default conv init, which is
      what the reference Painter keeps: ``create_generator`` never calls ``init_weights`` on it,
      reference generator.py:30-58, and spectral-norm-wrapped convs have no ``weight`` attr at init time,
      tutils.py:58-60)
    * biases: same bound (needs fan_in -> taken from the sibling weight's shape)
    * ``weight_u`` / ``weight_v``: l2-normalised random vectors (reference norms.py:129-133)
    * BatchNorm ``weight``: 1 + 0.1 U(-1,1); ``bias``: 0.1 U(-1,1); ``running_mean``: 0.1 U(-1,1);
      ``running_var``: 1 + 0.2 U(0,1); ``num_batches_tracked``: 0
    ``gain`` scales the conv-weight bound (the ResNet-101 fixtures use 1.6 so that 33 residual blocks neither
    vanish nor overflow fp16 with untrained weights).  ``res_gamma`` scales the LAST BatchNorm weight of every ResNet
    bottleneck (``*.bn3.weight``): values around 0.1-0.3 give the well-conditioned residual stack of a trained (or
    zero-gamma initialised) ResNet, whose gradients keep their direction under 16-bit storage -- unlike gain-1.6
    untrained weights with unit gammas, where each of the 33 blocks doubles the signal and the gradient is chaotic.
    """
    out = {}
    fan_in = {}
    for k, shp in shapes.items():
        if (k.endswith("weight") or k.endswith("weight_bar")) and len(shp) == 4:
            fan_in[k.rsplit(".", 1)[0]] = shp[1] * shp[2] * shp[3]
    for k, shp in shapes.items():
        shp = tuple(shp)
        s = key_seed(k, seed)
        base, leaf = k.rsplit(".", 1) if "." in k else ("", k)
        if leaf in ("weight", "weight_bar") and len(shp) == 4:
            b = gain / np.sqrt(fan_in[base])
            out[k] = uniform(shp, s, -b, b)
        elif leaf == "bias" and base in fan_in:
            b = 1.0 / np.sqrt(fan_in[base])
            out[k] = uniform(shp, s, -b, b)
        elif leaf in ("weight_u", "weight_v"):
            v = uniform(shp, s).astype(np.float64)
            out[k] = (v / (np.linalg.norm(v) + 1e-12)).astype(np.float32)
        elif leaf == "weight":  # norm scale
            out[k] = (1.0 + 0.1 * uniform(shp, s)).astype(np.float32)
            if res_gamma != 1.0 and base.endswith(".bn3"):
                out[k] = (out[k] * res_gamma).astype(np.float32)
        elif leaf == "bias":
            out[k] = (0.1 * uniform(shp, s)).astype(np.float32)
        elif leaf == "running_mean":
            out[k] = (0.1 * uniform(shp, s)).astype(np.float32)
        elif leaf == "running_var":
            out[k] = (1.0 + 0.2 * uniform01(shp, s)).astype(np.float32)
        elif leaf == "num_batches_tracked":
            out[k] = np.zeros(shp, dtype=np.int64)
        else:
            raise KeyError("fill_state_dict: no rule for key %r" % k)
    return out
