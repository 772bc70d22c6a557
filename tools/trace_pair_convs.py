"""Per-call durations of the split-precision convolutions of one apply_events batch (hybrid mode: split24 Masker, fp16
Painter): which layers still run on the gather kernel.  usage (GPU box): python tools/trace_pair_convs.py"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from climategan_amd import _lib, fill  # noqa: E402
from climategan_amd.config import default_opts  # noqa: E402
from climategan_amd.trainer import Trainer  # noqa: E402

dev = torch.device("cuda:0")
opts = default_opts()
opts.tasks = ["d", "s", "m", "p"]
T = Trainer(opts, device=dev).setup(inference=True)
shapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
T.G.load_state_dict({k: torch.from_numpy(v) for k, v in fill.fill_state_dict(shapes, seed=0, **bench.WELL_CONDITIONED).items()})
T.G.eval()
T.G.set_compute_dtype(sys.argv[1] if len(sys.argv) > 1 else "split24")
T.G.set_painter_compute_dtype(torch.float16)
T.overlap_branches = False
x = torch.from_numpy(fill.uniform((16, 3, 640, 640), 3000)).to(dev)
lib = _lib.load()
orig = lib.cgan_conv2d_nhwc_fwd_pair
rows = []


def traced(x3, w3, b, r, y3, dref, stream):
    d = dref._obj
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = orig(x3, w3, b, r, y3, dref, stream)
    e1.record()
    rows.append((e0, e1, "n%d %dx%d c%d -> %dx%d c%d k%d s%d d%d%s" % (d.n, d.h_in, d.w_in, d.c_in, d.h_out, d.w_out, d.c_out, d.kh,
                                                                     d.stride, d.dilation, " ups" if d.in_upsample else "")))
    return rc


for _ in range(2):
    T.infer_all(x, numpy=True, bin_value=0.5, half=False)
lib.cgan_conv2d_nhwc_fwd_pair = traced
T.infer_all(x, numpy=True, bin_value=0.5, half=False)
torch.cuda.synchronize()
lib.cgan_conv2d_nhwc_fwd_pair = orig
agg = {}
for e0, e1, tag in rows:
    a = agg.setdefault(tag, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(a[1] for a in agg.values())
print("split convs of one batch: %d calls, %.1f ms" % (len(rows), tot))
for tag, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    fl = None
    print("%7.3f ms %3d x  %s" % (ms, n, tag))
