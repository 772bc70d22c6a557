"""Run-to-run determinism of the joint train step: from identical states (parameters, optimizer state, RNG) the step leaves
bit-identical parameters, BatchNorm buffers and spectral-norm vectors -- on one stream, on two streams (the Masker and the
Painter branch of an update overlap on two HIP streams by default), and between the two.  Every reduction on the gradient
path sums in a fixed order (weight / bias gradients through the partial-tile workspace, norm backward sums through chunk rows,
the SIGM loss's statistics and Sobel gradient without atomics); only the logged loss SCALARS are accumulated with fp32 atomics
(they feed nothing)."""
import random
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def _run_steps(T, batch, sd_g, sd_d, overlap, steps=2):
    from climategan_amd import ops

    T.G.load_state_dict(sd_g)
    T.D.load_state_dict(sd_d)
    ops.touch(*T.G.parameters(), *T.G.buffers(), *T.D.parameters(), *T.D.buffers())
    T.g_opt.state.clear()
    T.d_opt.state.clear()
    T.global_step = 0
    T.overlap_branches = overlap
    random.seed(0)
    torch.manual_seed(0)
    for _ in range(steps):                      # an extrapolation and a real step of both optimizers
        T.train_step(batch)
    torch.cuda.synchronize()
    out = {"G." + k: v.clone() for k, v in T.G.state_dict().items()}
    out.update({"D." + k: v.clone() for k, v in T.D.state_dict().items()})
    return out


def test_train_step_is_bitwise_reproducible_on_one_and_on_two_streams():
    import bench

    dev = torch.device("cuda:0")
    T = bench.build_trainer(dev, torch.bfloat16)
    T.G.painter.set_latent_shape((2, 3, bench.H, bench.W), True)
    batch = bench.joint_batch(2, 0, dev)
    sd_g = {k: v.clone() for k, v in T.G.state_dict().items()}
    sd_d = {k: v.clone() for k, v in T.D.state_dict().items()}
    runs = [("one stream", False), ("one stream again", False), ("two streams", True), ("two streams again", True)]
    res = {name: _run_steps(T, batch, sd_g, sd_d, ov) for name, ov in runs}
    base = res["one stream"]
    moved = sum(not torch.equal(base[k], (sd_g if k[0] == "G" else sd_d)[k[2:]]) for k in base)
    assert moved > 1000, moved                                  # the steps did train
    for name in ("one stream again", "two streams", "two streams again"):
        bad = [k for k in base if not torch.equal(base[k], res[name][k])]
        detail = ["%s (max rel diff %.2g)" % (k, ((base[k].float() - res[name][k].float()).abs().max()
                                                   / (base[k].float().abs().max() + 1e-30)).item()) for k in bad[:12]]
        assert not bad, "%s: %d of %d tensors differ from the first one-stream run: %s" % (name, len(bad), len(base), detail)
