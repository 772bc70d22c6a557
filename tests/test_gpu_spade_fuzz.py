"""Seeded random-shape sweep of the fused SPADE kernel (the headline kernel) against the torch fp32 composition of
tests/test_gpu_ops.py::test_spade_fused: ragged image sizes (tiles cut by the border on both axes, images smaller than a
tile), channel counts that leave partial channel tiles / partial chunks, conditioning maps at other resolutions (legacy
nearest resize up and down), the folded x2 upsample, both activations."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def draw(n, seed):
    rng = np.random.RandomState(seed)
    cases = []
    for _ in range(n):
        ups = bool(rng.rand() < 0.3)
        H = int(rng.choice([2, 6, 10, 16, 18, 30, 32, 34, 50]))
        W = int(rng.choice([2, 4, 14, 16, 20, 32, 36, 48, 66]))
        if ups:
            H, W = H + H % 2, W + W % 2
        C = int(rng.choice([1, 5, 8, 12, 20, 33, 40, 41, 56, 80, 96, 100]))
        ch = int(rng.choice([H, 2 * H, max(H // 2, 1), H + 3]))
        cw = int(rng.choice([W, 2 * W, max(W // 2, 1), W + 5]))
        cases.append((C, H, W, (ch, cw), ups, str(rng.choice(["none", "lrelu"]))))
    return cases


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", draw(24, 4242))
def test_spade_fused_random_shapes(dt, case):
    from test_gpu_ops import test_spade_fused
    test_spade_fused(dt, case)
