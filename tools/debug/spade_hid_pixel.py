"""debug: where does hidden row 82's mask differ?"""
import sys
from pathlib import Path
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from climategan_amd import fill, ops


def q(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt).float()


dt = torch.float16
C, H, W, B = 5, 112, 96, 3
dgb_f = q(fill.uniform((B, 2 * C, H, W), 7100 + C, -1, 1), dt)
seg_f = q(fill.uniform((B, 3, H, W), 7101 + H), dt)
w_sh = q(fill.uniform((128, 3, 3, 3), 7102, -0.4, 0.4), dt)
b_sh = torch.from_numpy(fill.uniform((128,), 7103, -0.2, 0.2))
w_gb = q(fill.uniform((2 * C, 128, 3, 3), 7104 + C, -0.05, 0.05), dt)
dgb = ops.nchw_to_nhwc(dgb_f.cuda(), dt)
seg = ops.nchw_to_nhwc(seg_f.cuda(), dt)
pw = ops.pack_conv_weight(w_sh.cuda(), b_sh.cuda(), dt)
a = ops.conv2d(seg, pw, pad=1, act=ops.ACT_RELU)
h64 = F.conv2d(seg_f.double(), w_sh.double(), b_sh.double(), padding=1)          # float64 reference of h
a_nchw = ops.nhwc_to_nchw(a).cpu()
hid = 82
m_u = a_nchw[:, hid] > 0
m_t = h64[:, hid] > 0
diff = (m_u != m_t).nonzero()
print("pixels where the stored map's mask differs from float64's:", diff.tolist()[:10])
small = (h64[:, hid].abs() < 1e-4).nonzero()
print("pixels with |h| < 1e-4:", [(tuple(i.tolist()), h64[:, hid][tuple(i.tolist())].item(), a_nchw[:, hid][tuple(i.tolist())].item()) for i in small[:10]])
d_pre = ops.conv2d_bwd_data(dgb, w_gb.cuda(), (B, H, W), pad=1, relu_out=a)
dw, db = ops.spade_hidden_bwd(dgb, w_gb.cuda(), seg, pw, C)
dw_u, db_u = ops.conv2d_bwd_weight(seg, d_pre, (128, 3, 3, 3), pad=1)
print("db[82] fused %.6f unfused %.6f diff %.6f" % (db[hid].item(), db_u[hid].item(), (db[hid] - db_u[hid]).item()))
# unmasked dh at the candidate pixels
d_raw = ops.conv2d_bwd_data(dgb, w_gb.cuda(), (B, H, W), pad=1)
dr = ops.nhwc_to_nchw(d_raw).cpu()
for i in small[:10]:
    t = tuple(i.tolist())
    print("   ", t, "unmasked dh", dr[:, hid][t].item())
