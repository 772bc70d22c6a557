"""Dev tool (GPU box): per-block error of the HIP Painter vs the CPU oracle on a golden case."""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from helpers import case_state_dict, golden_cases, t  # noqa: E402
from oracle import cpu_ref  # noqa: E402
from oracle.make_golden import case_inputs  # noqa: E402

from climategan_amd import ops  # noqa: E402
from climategan_amd.config import default_opts  # noqa: E402
from climategan_amd.generator import create_generator  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "painter_up7"
dt = torch.float16 if (len(sys.argv) < 3 or sys.argv[2] == "fp16") else torch.bfloat16
case = golden_cases()[name]
opts = default_opts(); opts.tasks = ["p"]
opts.gen.p.latent_dim = case["latent_dim"]; opts.gen.p.spade_n_up = case["n_up"]
G = create_generator(opts, device="cuda")
sd = case_state_dict(case)
G.painter.load_state_dict(sd)
G.set_compute_dtype(dt)
P = G.painter
H, W = case["H"], case["W"]
zh, zw = H // 2 ** case["n_up"], W // 2 ** case["n_up"]
cond = t(case_inputs(name, case)["cond"])
cond_g = ops.nchw_to_nhwc(cond.cuda(), dt, cs=4)

def rep(tag, got, ref):
    e = (got - ref).abs()
    idx = e.flatten().argmax().item()
    pos = []
    for s in reversed(ref.shape):
        pos.append(idx % s); idx //= s
    print("%-28s shape %-20s max err %.4g mean %.4g at %s ; ref absmax %.3g" % (
        tag, tuple(ref.shape), e.max().item(), e.mean().item(), tuple(reversed(pos)), ref.abs().max().item()))

with torch.no_grad():
    # oracle chain
    z = F.conv2d(cpu_ref.nearest_resize(cond, (zh, zw)), sd["fc.weight"], sd["fc.bias"], padding=1)
    zg = P.forward_nhwc  # noqa
    zin = ops.resize_nearest(cond_g, (zh, zw), cs_out=8)
    from climategan_amd.norms import conv_forward
    y_g = conv_forward(P.fc, P._fc_cache, zin)
    rep("fc", ops.nhwc_to_nchw(y_g).cpu(), z)
    y = z
    blocks = [("head_0", P.head_0, False), ("G_middle_0", P.G_middle_0, True), ("G_middle_1", P.G_middle_1, True)]
    blocks += [("up_spades.%d" % i, b, True) for i, b in enumerate(P.up_spades)]
    blocks += [("final_spade", P.final_spade, False)]
    up = lambda a: cpu_ref.nearest_resize(a, (a.shape[-2] * 2, a.shape[-1] * 2))
    G2 = create_generator(opts, device="cuda"); G2.painter.load_state_dict(case_state_dict(case)); G2.set_compute_dtype(dt)
    P2 = G2.painter
    blocks2 = [P2.head_0, P2.G_middle_0, P2.G_middle_1] + list(P2.up_spades) + [P2.final_spade]
    chain = conv_forward(P2.fc, P2._fc_cache, zin)
    for (bname, blk, ups), blk2 in zip(blocks, blocks2):
        y_in = y
        chain = blk2.forward_nhwc(chain, cond_g, x_upsample=ups, post_act="lrelu" if bname == "final_spade" else None)
        # isolated: HIP block fed with the oracle's input (pre-upsample), fresh u/v
        xin = ops.nchw_to_nhwc(y_in.cuda(), dt)
        iso = blk.forward_nhwc(xin, cond_g, x_upsample=ups)
        # chained HIP (u/v already advanced by the isolated call -> reload state first)
        yy = up(y) if ups else y
        y = cpu_ref.spade_resnet_block(yy, cond, sd, bname)
        rep(bname + " (isolated)", ops.nhwc_to_nchw(iso).cpu(), y)
        rep(bname + " (chained)", ops.nhwc_to_nchw(chain).cpu(), F.leaky_relu(y, 0.2) if bname == "final_spade" else y)
    out = conv_forward(P2.conv_img, P2._img_cache, chain, act=ops.ACT_TANH)
    ref_out = torch.tanh(F.conv2d(F.leaky_relu(y, 0.2), sd["conv_img.weight"], sd["conv_img.bias"], padding=1))
    rep("output (chained)", ops.nhwc_to_nchw(out).cpu(), ref_out)
    # emulation: oracle with activations rounded to the compute dtype after every block
