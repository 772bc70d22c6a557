"""Dev-container measurement (needs /root/reference): the REFERENCE's own ``Trainer.update_G`` on the jstep fixtures with
every Conv2d / BatchNorm2d / InstanceNorm2d / ReLU / LeakyReLU / Tanh output of G -- and the gradient flowing back
through it -- rounded to a 16-bit type (what a 16-bit-storage training path does), vs its fp32 run: per parameter group
the cosine and norm ratio of the gradients.  This is the yardstick for the direction asserts of
tests/test_gpu_configs_640.py.   usage: python tests/devtools/measure_ref_jstep_quant.py <jstep_small|jstep_640> [bf16|fp16]"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import oracle.make_golden_640 as M
from helpers import t

name = sys.argv[1]
qdt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.bfloat16
case = M.CASES_640[name]
torch.set_num_threads(8)


def run(quant):
    T = M.reference_training_trainer(case)
    if quant:
        rq = lambda v: v.to(qdt).float() if torch.is_tensor(v) and v.is_floating_point() else v
        kinds = (torch.nn.Conv2d, torch.nn.BatchNorm2d, torch.nn.InstanceNorm2d, torch.nn.ReLU, torch.nn.LeakyReLU, torch.nn.Tanh)
        for root in (T.G, T.D, T.losses["G"]["p"]["vgg"]):
            for mod in root.modules():
                if isinstance(mod, (torch.nn.ReLU, torch.nn.LeakyReLU)):
                    mod.inplace = False
                if isinstance(mod, kinds):
                    def fwd_hook(m, i, o):
                        o = rq(o).clone()
                        if o.requires_grad:
                            o.register_hook(rq)
                        return o
                    mod.register_forward_hook(fwd_hook)
    batch = {dom: {"data": {k: t(v) for k, v in d.items()}} for dom, d in M.jstep_inputs(case).items()}
    saved = (torch.Tensor.cuda, torch.Tensor.get_device)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.get_device = lambda self: torch.device("cpu")
    try:
        T.update_G(batch)
        out = {k: p.grad.clone() for k, p in T.G.named_parameters() if p.requires_grad and p.grad is not None}
        T.update_D(batch)          # on the generator ExtraAdam has just extrapolated with those gradients
        out.update({"D." + k: p.grad.clone() for k, p in T.D.named_parameters() if p.requires_grad and p.grad is not None})
    finally:
        torch.Tensor.cuda, torch.Tensor.get_device = saved
    return out


g32, gq = run(False), run(True)
groups = {}
for k in g32:
    a, b = g32[k].flatten().double(), gq[k].flatten().double()
    if a.norm() == 0:
        continue
    base = k.rsplit(".", 1)[0]
    sib = max([float(g32[n].norm()) for n in (base + ".weight", base + ".weight_bar") if n in g32] + [0.0])
    if float(a.norm()) < 1e-4 * max(sib, 1e-2):
        continue                                       # zero-true-gradient biases in front of a norm layer
    c = float((a * b).sum() / (a.norm() * b.norm() + 1e-300))
    is_conv = k.endswith("weight_bar") or (k.endswith(".weight") and g32[k].dim() == 4)
    grp = ("encoder conv" if is_conv else "encoder bn") if k.startswith("encoder.") else k.split(".")[0]
    if k.startswith("D."):
        grp = "D." + k.split(".")[1]
    groups.setdefault(grp, []).append((c, float(b.norm() / a.norm()), k))
print(name, qdt)
for grp, v in groups.items():
    cs, rs = np.array([x[0] for x in v]), np.array([x[1] for x in v])
    print("%-13s n=%4d  norm ratio median %.3f [%.3f, %.3f]   cos median %.4f p10 %.4f min %.4f  (%s)" % (
        grp, len(v), np.median(rs), rs.min(), rs.max(), np.median(cs), np.percentile(cs, 10), cs.min(), min(v)[2]))
