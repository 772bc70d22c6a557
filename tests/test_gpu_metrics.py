"""Validation metrics on the device against the reference's own ``accuracy`` / ``mIOU``
(eval_metrics.py:67-130), whose outputs for seeded inputs are committed in tests/golden/metrics.json
(oracle/make_golden_metrics.py).  Integer counting: the ratios must be exactly the reference's."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from climategan_amd import fill

pytestmark = pytest.mark.gpu
GOLD = json.loads((Path(__file__).resolve().parent / "golden" / "metrics.json").read_text())


def inputs(case):
    n, c, h, w, seed = case["n"], case["c"], case["h"], case["w"], case["seed"]
    logits = torch.from_numpy(fill.uniform((n, c, h, w), seed, -3, 3)).half().float()    # exactly representable in fp16
    label = torch.from_numpy((fill.uniform01((n, 1, h, w), seed + 1) * (c + case["extra"])).astype(np.int64))
    return logits, (label[:, 0] if case["label_dims"] == 3 else label)


@pytest.mark.parametrize("name", sorted(GOLD["seg"]))
def test_seg_metrics_match_reference(name):
    from climategan_amd import eval_metrics, ops
    case = GOLD["seg"][name]
    logits, label = inputs(case)
    for pred in (logits.cuda(), ops.nchw_to_nhwc(logits.cuda(), torch.float16)):
        assert eval_metrics.accuracy(pred, label.cuda()) == case["accuracy"]
        for avg in ("macro", "weighted"):
            got, ref = eval_metrics.mIOU(pred, label.cuda(), avg), case["mIOU_" + avg]
            assert (np.isnan(got) and ref is None) or got == pytest.approx(ref, rel=1e-15, abs=0)


def test_mask_metrics_match_reference():
    """eval_images' mask branch (trainer.py:1759-1772): accuracy(pred_mask, m) on same-rank maps -- which in the
    reference is the fraction of ZERO labels, because a 1-channel prediction is arg-maxed to all zeros
    (eval_metrics.py:73-75) -- and mIOU(cat[1-p, p], m)."""
    from climategan_amd import eval_metrics
    case = GOLD["mask"]
    p = torch.from_numpy((fill.uniform01((case["n"], 1, case["h"], case["w"]), case["seed"]) > 0.5).astype(np.float32))
    m = torch.from_numpy((fill.uniform01((case["n"], 1, case["h"], case["w"]), case["seed"] + 1) > 0.4).astype(np.float32))
    assert eval_metrics.accuracy(p.cuda(), m.cuda()) == case["accuracy"]
    prob = torch.cat([1 - p, p], dim=1)
    assert eval_metrics.mIOU(prob.cuda(), m.cuda()) == pytest.approx(case["mIOU"], rel=1e-15, abs=0)
    with pytest.raises(NotImplementedError, match="broadcast"):
        eval_metrics.accuracy(torch.cat([prob, prob]).cuda(), torch.cat([m, m]).cuda())
