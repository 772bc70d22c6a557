import sys, random
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from helpers import load_golden, t
from oracle.make_golden_640 import CASES_640, generator_fill, infer_inputs
from climategan_amd.config import default_opts
from climategan_amd.trainer import Trainer
case = CASES_640["infer_640"]
opts = default_opts(); opts.tasks = ["d","s","m","p"]
T = Trainer(opts, device="cuda").setup(inference=True)
shapes = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
sd = generator_fill(shapes, case)
T.G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
T.G.eval(); T.G.float()
gold = load_golden("infer_640")
B,H,W = 2,640,640
x2 = t(infer_inputs(case)["x"]).cuda()
with torch.no_grad():
    m = T.G.masker_forward(x2, sigmoid=False)["m"]
ref_mask = np.unpackbits(gold["mask_bits"])[: B*H*W].reshape(B,1,H,W).astype(bool)
logit = m.cpu().numpy()
got = logit > 0
idx = np.argwhere(got != ref_mask)
print("differing pixels:", idx.tolist())
for i in idx: print("  logit there:", logit[tuple(i)], "ref bit", ref_mask[tuple(i)])
a = np.abs(logit).ravel(); a.sort()
print("smallest |logit| values:", a[:8])
