"""How far is the fp32 CPU evaluation of the eval-mode Masker from float64 on the infer_640 fixture?  (the noise floor any
fp32-grade implementation sits at when compared with the reference's fp32 golden)"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from oracle import cpu_ref
from oracle.make_golden_640 import CASES_640, generator_fill, infer_inputs
import json
case = CASES_640["infer_640"]
shapes = {k: tuple(v) for k, v in json.load(open('/root/repo/tests/golden/generator_masker_shapes.json')).items()}
sd = {k: torch.from_numpy(v) for k, v in generator_fill(shapes, case).items()}
x = torch.from_numpy(infer_inputs(case)["x"])[:1]
torch.set_num_threads(8)
outs = {}
for dt in (torch.float32, torch.float64):
    t0 = time.time()
    s = {k: v.to(dt).clone() for k, v in sd.items()}
    with torch.no_grad():
        o = cpu_ref.masker_forward(s, x.to(dt), (160, 160), update=True, d_target=160)
    outs[dt] = o
    print(dt, "%.1f s" % (time.time() - t0), type(o), (list(o.keys()) if isinstance(o, dict) else len(o)))
a, b = outs[torch.float32], outs[torch.float64]
items = a.items() if isinstance(a, dict) else enumerate(a)
for k, v in items:
    w = b[k]
    if torch.is_tensor(v):
        err = (v.double() - w).abs().max().item(); sc = w.abs().max().item()
        print(k, "max |fp32 - fp64| = %.3g, scale %.3g, relative %.2g" % (err, sc, err / sc))
