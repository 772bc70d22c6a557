"""debug: fused vs unfused hidden-map backward over a sweep of shapes"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import fill, ops


def q(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).float()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


for dt in (torch.float16, torch.bfloat16):
    for C, H, W, B in [(5, 112, 96, 1), (5, 112, 96, 2), (5, 112, 96, 3), (5, 96, 96, 3), (5, 80, 80, 3), (20, 112, 96, 3), (5, 112, 96, 4),
                       (8, 112, 96, 3), (4, 112, 96, 3)]:
        dgb_f = q(fill.uniform((B, 2 * C, H, W), 7100 + C, -1, 1), dt)
        seg_f = q(fill.uniform((B, 3, H, W), 7101 + H), dt)
        w_sh = q(fill.uniform((128, 3, 3, 3), 7102, -0.4, 0.4), dt)
        b_sh = torch.from_numpy(fill.uniform((128,), 7103, -0.2, 0.2))
        w_gb = q(fill.uniform((2 * C, 128, 3, 3), 7104 + C, -0.05, 0.05), dt)
        dgb = ops.nchw_to_nhwc(dgb_f.cuda(), dt)
        seg = ops.nchw_to_nhwc(seg_f.cuda(), dt)
        pw = ops.pack_conv_weight(w_sh.cuda(), b_sh.cuda(), dt)
        dw, db = ops.spade_hidden_bwd(dgb, w_gb.cuda(), seg, pw, C)
        a = ops.conv2d(seg, pw, pad=1, act=ops.ACT_RELU)
        d_pre = ops.conv2d_bwd_data(dgb, w_gb.cuda(), (B, H, W), pad=1, relu_out=a)
        dw_u, db_u = ops.conv2d_bwd_weight(seg, d_pre, (128, 3, 3, 3), pad=1)
        db_t = d_pre.t.float().sum((0, 1, 2))[:128]
        bad = ((db - db_u).abs() > 1e-3 * db_u.abs().max()).nonzero().flatten().tolist()
        print(dt, C, H, W, B, "dw %.2e db %.2e  db_u vs sum(d_pre) %.2e  bad hidden rows %s" % (rel(dw, dw_u), rel(db, db_u), rel(db_u, db_t), bad[:12]),
              flush=True)
