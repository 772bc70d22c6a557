import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from helpers import golden_cases, load_golden
from climategan_amd import fill
import test_gpu_train as TT
case = golden_cases()["mstep"]; gold = load_golden("mstep")
dt = torch.bfloat16 if "fp16" not in sys.argv else torch.float16
from climategan_amd import autograd as AG
S = 8192.0 if "fp16" in sys.argv else 1.0
AG.set_grad_scale(S)
T = TT.build_masker_trainer(case, dt)
for p in T.D.parameters(): p.requires_grad_(False)
loss = T.get_masker_loss(TT.masker_batch(case)); loss.backward()
params = dict(T.G.named_parameters())
rows = []
for gk in gold:
    if not gk.startswith("gsub."): continue
    key = gk[5:]; g = params[key].grad
    flat = g.reshape(-1).float().cpu().numpy() / S; ref = gold[gk].astype(np.float64); n = case["sub"]
    if flat.size > n:
        idx = (fill.uniform01((n,), fill.key_seed(key, 4242)) * flat.size).astype(np.int64).clip(0, flat.size - 1); sub = flat[idx].astype(np.float64)
    else: sub = flat.astype(np.float64)
    rn = float(gold["gnorm." + key][0])
    cos = (sub * ref).sum() / max(np.sqrt((sub ** 2).sum() * (ref ** 2).sum()), 1e-30)
    rows.append((key, cos, float(np.linalg.norm(flat)) / max(rn, 1e-30), rn))
sel = [r for r in rows if r[0].endswith("weight") or r[0].endswith("weight_bar")]
print("loss", loss.item(), float(gold["loss"][0]))
for r in sel[:4] + sel[60:62] + sel[150:152] + sel[230:232] + sel[-34:-28] + sel[-20:-14] + sel[-3:]:
    print("%-60s cos %7.4f ratio %8.3f refnorm %.3e" % r)
