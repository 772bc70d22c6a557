import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
from helpers import golden_cases, load_golden, t
from oracle.make_golden import case_inputs
import test_gpu_train as T
case = golden_cases()["dstep_p"]; gold = load_golden("dstep_p")
from climategan_amd import autograd as AG
for dt, S in ((torch.float16, 1.0), (torch.float16, 256.0), (torch.float16, 4096.0), (torch.bfloat16, 1.0)):
    AG.set_grad_scale(S)
    D = T.build_D(case, dt)
    inp = {k: t(v).cuda() for k, v in case_inputs("dstep_p", case).items()}
    loss = T.d_loss(D, inp); loss.backward()
    print(dt, "S", S, "loss", loss.item(), float(gold["loss"][0]))
    for key, p in D.named_parameters():
        if not p.requires_grad or not key.endswith("weight_bar"): continue
        ref = gold["grad." + key]; got = p.grad.cpu().numpy() / S
        err = np.abs(got - ref); i = np.unravel_index(err.argmax(), err.shape)
        cos = (got * ref).sum() / np.sqrt((got**2).sum() * (ref**2).sum())
        print("%-45s scale %.2e max %.2e at %s ref %.3e got %.3e  rel-l2 %.3e cos %.6f" % (key, np.abs(ref).max(), err.max(), i, ref[i], got[i], np.sqrt((err**2).sum()/(ref**2).sum()), cos))
