"""tests/golden/metrics.json: outputs of the REFERENCE's own ``accuracy`` / ``mIOU`` (climategan/eval_metrics.py:67-130)
on seeded inputs (dev container only; test infrastructure).   python -m oracle.make_golden_metrics"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

from climategan_amd import fill

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "metrics.json"
# label_dims 4 = [N, 1, H, W] as eval_images passes them (batch 1: with a larger batch the reference's un-squeezed ``gt``
# broadcasts against the argmax, eval_metrics.py:68-76); label_dims 3 = [N, H, W], any batch
SEG = {"seg11": dict(n=1, c=11, h=40, w=56, seed=300, extra=0, label_dims=4),
       "seg11_batch": dict(n=3, c=11, h=24, w=40, seed=305, extra=0, label_dims=3),
       "seg11_ignore": dict(n=1, c=11, h=33, w=47, seed=310, extra=2, label_dims=4),   # labels 11, 12: ignore index and beyond
       "seg19_absent": dict(n=1, c=19, h=16, w=16, seed=320, extra=-12, label_dims=4)}  # classes 7..18 never labelled


def main():
    from oracle import ref_shim

    if not ref_shim.available():
        sys.exit("needs /root/reference (dev container only)")
    em = ref_shim.ref("eval_metrics")
    out = {"seg": {}, "mask": None}
    for name, case in SEG.items():
        n, c, h, w, seed = case["n"], case["c"], case["h"], case["w"], case["seed"]
        logits = torch.from_numpy(fill.uniform((n, c, h, w), seed, -3, 3)).half().float()
        label = torch.from_numpy((fill.uniform01((n, 1, h, w), seed + 1) * (c + case["extra"])).astype(np.int64))
        if case["label_dims"] == 3:
            label = label[:, 0]
        res = dict(case)
        res["accuracy"] = em.accuracy(logits, label)
        for avg in ("macro", "weighted"):
            v = float(em.mIOU(logits, label, avg))
            res["mIOU_" + avg] = None if np.isnan(v) else v
        out["seg"][name] = res
    mc = dict(n=1, h=48, w=64, seed=400)
    p = torch.from_numpy((fill.uniform01((mc["n"], 1, mc["h"], mc["w"]), mc["seed"]) > 0.5).astype(np.float32))
    m = torch.from_numpy((fill.uniform01((mc["n"], 1, mc["h"], mc["w"]), mc["seed"] + 1) > 0.4).astype(np.float32))
    mc["accuracy"] = em.accuracy(p, m)
    mc["mIOU"] = float(em.mIOU(torch.cat([1 - p, p], dim=1), m))
    out["mask"] = mc
    OUT.write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
