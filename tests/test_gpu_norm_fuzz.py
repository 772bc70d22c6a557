"""Seeded random-shape sweep of the normalisation kernels (instance norm and training-mode batch norm, forward and
backward, with their activations) through tests/test_gpu_backward.py's torch-autograd comparisons: channel counts that
need padding or exceed one workgroup's channel span (> 2048 storage channels are split), single-row and single-column
maps, very few pixels per channel (chunked reductions with empty tails)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def draw(n, seed):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        c = int(rng.choice([1, 3, 8, 20, 64, 100, 256, 1000, 2048]))
        h = int(rng.choice([1, 2, 5, 9, 16, 33]))
        w = int(rng.choice([2, 3, 8, 17, 40]))
        if h * w < 4:
            w = 4
        if c * h * w > 2_000_000:
            h, w = 5, 8
        out.append((c, h, w))
    return out


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act", ["none", "lrelu"])
@pytest.mark.parametrize("shape", draw(10, 77))
def test_instance_norm_random_shapes(dt, act, shape):
    from test_gpu_backward import test_instnorm_act_backward
    c, h, w = shape
    test_instnorm_act_backward(dt, c, h, w, act)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("act,affine", [("relu", True), ("none", True), ("lrelu", False)])
@pytest.mark.parametrize("shape", draw(8, 78))
def test_batch_norm_random_shapes(dt, act, affine, shape):
    from test_gpu_backward import test_batchnorm_training_forward_backward
    c, h, w = shape
    test_batchnorm_training_forward_backward(dt, c, h, w, act, affine)
