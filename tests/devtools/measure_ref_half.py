"""Dev-container tool: how far does the REFERENCE's own half-precision path (module.to(fp16/bf16), CPU) deviate
from its fp32 output on the painter fixtures?  The numbers are the yardstick for the end-to-end 16-bit tolerance in
tests/test_gpu_painter.py (REF_HALF_DEV).  Needs /root/reference."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import load_golden, t  # noqa: E402
from oracle.make_golden import build_reference_module, case_inputs, golden_cases, summarize  # noqa: E402

cases = golden_cases()
for name in ["painter_up4", "painter_up7", "painter_640", "paint_up4"]:
    case = cases[name]
    gold = load_golden(name)
    for dt in (torch.float16, torch.bfloat16):
        mod, _ = build_reference_module(case)
        mod = mod.to(dt)
        inp = {k: t(v).to(dt) for k, v in case_inputs(name, case).items()}
        x = inp.get("cond", inp.get("x"))
        mod.set_latent_shape(tuple(x.shape), True)
        with torch.no_grad():
            if case["kind"] == "paint":
                m = inp["m"]
                y = (x * (1 - m) + mod(None, x * (1 - m)) * m).float().numpy()
            else:
                y = mod(None, x).float().numpy()
        if case["full"]:
            e = np.abs(y - gold["y"])
            mx, mn = e.max(), e.mean()
        else:
            s = summarize(y)
            es = [np.abs(s[k] - gold["y_" + k]) for k in ("crop_tl", "crop_c", "crop_br")]
            mx, mn = max(e.max() for e in es), max(e.mean() for e in es)
        print('    ("%s", "%s"): (%.4g, %.4g),' % (name, str(dt).split(".")[1], mx, mn))
