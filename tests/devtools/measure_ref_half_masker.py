"""Dev-container tool (needs /root/reference): the REFERENCE's own 16-bit Masker at 640 x 640 -- ``G.half()`` /
``G.bfloat16()`` on the CPU -- vs its fp32 run on the ``infer_640`` fixture; prints the (max, mean) deviation of the depth
/ segmentation / mask outputs relative to each output's scale, in the form of tests/test_gpu_configs_640.py's
REF_HALF_MASKER table.  usage: python tests/devtools/measure_ref_half_masker.py [fp16] [bf16]"""
import contextlib, io, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from helpers import load_golden, t
from oracle import ref_shim
from oracle.make_golden import summarize
import oracle.make_golden_640 as M

case = M.CASES_640["infer_640"]
gold = load_golden("infer_640")
opts = ref_shim.default_opts(); opts.tasks = ["d", "s", "m", "p"]
tr = ref_shim.ref("trainer"); tr.Timer = M._NullTimer
T = tr.Trainer(opts, device=torch.device("cpu"))
with contextlib.redirect_stdout(io.StringIO()):
    T.setup(inference=True)
shapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
sd = {k: t(v) for k, v in M.generator_fill(shapes, case).items()}
x = t(M.infer_inputs(case)["x"])
for name in [a for a in sys.argv[1:]] or ["bf16"]:
    dt = torch.float16 if name == "fp16" else torch.bfloat16
    T.G.float(); T.G.load_state_dict(sd); T.G.eval(); T.G.to(dt)
    with torch.no_grad():
        z = T.G.encode(x.to(dt)); d, zd = T.G.decoders["d"](z); s = T.G.decoders["s"](z, zd)
        m = T.G.mask(z=z, cond=None, z_depth=zd)
    for k, y in (("d", d), ("s", s), ("m", m)):
        su = summarize(y.float().numpy())
        scale = max(np.abs(gold[k + "_crop_c"]).max(), np.abs(gold[k + "_pooled8"]).max())
        errs = np.concatenate([np.abs(su[c] - gold["%s_%s" % (k, c)]).ravel() for c in ("crop_tl", "crop_c", "crop_br")])
        if k == "m":
            ref = np.concatenate([gold["m_" + c].ravel() for c in ("crop_tl", "crop_c", "crop_br")])
            errs = errs[np.abs(ref - 0.5) > 0.45]
        print('    ("%s", "%s"): (%.4g, %.4g),' % (k, "float16" if name == "fp16" else "bfloat16", errs.max() / scale, errs.mean() / scale), flush=True)
