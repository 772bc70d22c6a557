"""GPU-box diagnosis of the ADVENT discriminators' gradient direction in the joint step (jstep_small): where does the
deviation from the reference's fp32 step come from -- the discriminator's INPUT (the generator the G update produced)
or the discriminator's own 16-bit pipeline?

  mine    : Trainer.update_G + update_D on the HIP path (bf16)
  hybrid  : the oracle's fp32 discriminator (cpu_ref.fc_discriminator, this Trainer's D state right before update_D) on
            the logits / depth the HIP generator handed to the discriminator
  oracle  : cpu_ref.joint_train_step (fp32 everywhere, pinned by the reference's own step)

usage: python tests/devtools/diag_advent_d.py [jstep_small]"""
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_configs_640 as TG            # noqa: E402
from climategan_amd import fill, losses as L, ops   # noqa: E402
from helpers import t                          # noqa: E402
from oracle import cpu_ref                     # noqa: E402
from oracle.make_golden_640 import CASES_640, generator_fill, jstep_inputs   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "jstep_small"
case = CASES_640[name]
T = TG._build_train(("d", "s", "m", "p"), case, 1)
batch = TG._batch(case, 1, ("r", "s", "rf"))
T.G.painter.set_latent_shape((case["B"], 3, case["H"], case["W"]), True)
T.update_G(batch)
d_before = {k: v.detach().float().cpu().clone() for k, v in T.D.state_dict().items()}

calls = []
orig = L.advent_input


def spy(logits, depth=None, sigmoid_pair=False):
    calls.append((ops.nhwc_to_nchw(ops.NHWC(logits.t.detach(), logits.c)).float().cpu(),
                  None if depth is None else ops.nhwc_to_nchw(ops.NHWC(depth.t.detach(), depth.c)).float().cpu(),
                  sigmoid_pair))
    return orig(logits, depth, sigmoid_pair)


L.advent_input = spy
T.update_D(batch)
L.advent_input = orig
mine = {k: p.grad.detach().float().cpu() for k, p in T.D.named_parameters() if p.grad is not None}
print("captured discriminator calls:", [(tuple(c[0].shape), c[2]) for c in calls])

# hybrid: fp32 discriminator on the captured inputs, per task in call order (domain r first, then s)
def hybrid_grads(call_list):
    dd = cpu_ref._trainable(d_before)
    for k, v in dd.items():
        v.requires_grad_(k.rsplit(".", 1)[-1] not in ("weight_u", "weight_v"))
    total = 0
    seen = {"s": 0, "m": 0}
    for logits, depth, sig in call_list:
        task = "m" if sig else "s"
        label = [1.0, 0.0][seen[task]]            # r (label 1) then s (label 0): the trainer's domain order
        seen[task] += 1
        if sig:
            p = torch.sigmoid(logits)
            ent = cpu_ref.prob_2_entropy(torch.cat([p, 1 - p], 1))
        else:
            ent = cpu_ref.prob_2_entropy(torch.softmax(logits, 1)) * depth
        o = cpu_ref.fc_discriminator(ent, dd, task + ".Advent")
        total = total + F.binary_cross_entropy_with_logits(o, torch.full_like(o, label))
    ks = [k for k, v in dd.items() if v.requires_grad and (k.startswith("s.") or k.startswith("m."))]
    grads = torch.autograd.grad(total, [dd[k] for k in ks], allow_unused=True)
    return ks, {k: g for k, g in zip(ks, grads) if g is not None}


keys, hybrid = hybrid_grads(calls)

# oracle: the full fp32 step
gs = {k: tuple(v.shape) for k, v in T.G.state_dict().items()}
ds = {k: tuple(v.shape) for k, v in T.D.state_dict().items()}
sd_g = {k: t(v) for k, v in generator_fill(gs, case).items()}
sd_d = {k: t(v) for k, v in fill.fill_state_dict(ds, case["seed"] + 1).items()}
sd_v = {k: t(v) for k, v in fill.fill_state_dict(cpu_ref.vgg19_shapes(), case["vgg_seed"], gain=case["vgg_gain"]).items()}
cb = {dom: {k: t(v) for k, v in d.items()} for dom, d in jstep_inputs(case).items()}
torch.set_num_threads(32)
out = cpu_ref.joint_train_step(sd_g, sd_d, sd_v, cb, case.get("n_up", 7), 3, case.get("n_layers", 4))
oracle = out["d_grads"]


def cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a * b).sum() / (a.norm() * b.norm() + 1e-300))


print("%-34s %10s %10s | cos(mine,oracle) cos(mine,hybrid) cos(hybrid,oracle)" % ("tensor", "|oracle|", "|mine|"))
for k in keys:
    if k in mine and k in oracle and k in hybrid:
        print("%-34s %10.3g %10.3g | %8.4f %8.4f %8.4f" % (k, float(oracle[k].norm()), float(mine[k].norm()),
                                                         cos(mine[k], oracle[k]), cos(mine[k], hybrid[k]), cos(hybrid[k], oracle[k])))
# the D state the oracle had before its D update vs ours (spectral-norm vectors after the G-side forward)
for k in ("s.Advent.0.module.weight_u", "s.Advent.0.module.weight_v", "s.Advent.8.module.weight_u"):
    a = d_before[k]
    print(k, "norm", float(a.norm()))

# which input is it?  the oracle's own predictions at the D update (its extrapolated generator) vs the captured ones
g_o = out["g_state"]
pred_o = {}
with torch.no_grad(), cpu_ref.bn_training():
    for dom in ("r", "s"):
        b = cb[dom]
        pred_o[dom] = cpu_ref._masker_preds(g_o, b["x"], tuple(b["s"].shape[-2:]), b["d"].shape[-1])
s_calls = [c for c in calls if not c[2]]
for (lg, dp, _), dom in zip(s_calls, ("r", "s")):
    d_o, s_o, _ = pred_o[dom]
    print("domain %s: seg logits  max|mine - oracle| %.4g of scale %.4g (std over pixels %.4g);  depth max|diff| %.4g of scale %.4g (std %.4g)"
          % (dom, float((lg - s_o).abs().max()), float(s_o.abs().max()), float(s_o.std()),
             float((dp - d_o).abs().max()), float(d_o.abs().max()), float(d_o.std())))
    e_m = cpu_ref.prob_2_entropy(torch.softmax(lg, 1)) * dp
    e_o = cpu_ref.prob_2_entropy(torch.softmax(s_o, 1)) * d_o
    print("          D input   max|mine - oracle| %.4g of scale %.4g, std over pixels of the oracle's %.4g, of the difference %.4g"
          % (float((e_m - e_o).abs().max()), float(e_o.abs().max()), float(e_o.std()), float((e_m - e_o).std())))
m_calls = [c for c in calls if c[2]]
variants = {
    "oracle logits + oracle depth": [(pred_o[d][1], pred_o[d][0], False) for d in ("r", "s")],
    "oracle logits + my depth": [(pred_o[d][1], c[1], False) for d, c in zip(("r", "s"), s_calls)],
    "my logits + oracle depth": [(c[0], pred_o[d][0], False) for d, c in zip(("r", "s"), s_calls)],
}
for vn, cl in variants.items():
    _, hv = hybrid_grads(cl + m_calls)
    cs = [cos(hv[k], oracle[k]) for k in keys if k.startswith("s.") and k in hv and float(oracle[k].norm()) > 1e-3]
    print("%-32s cos(hybrid, oracle) over D.s tensors: median %.4f min %.4f" % (vn, sorted(cs)[len(cs) // 2], min(cs)))

# the discriminators' own 16-bit pipeline: which rounding costs the direction?  fp32 discriminator on the captured
# inputs with (a) conv WEIGHTS rounded to bf16 per call (what any bf16 MFMA path does: w_bar / sigma differs between the
# two domain calls, so the rounding differs too), (b) LeakyReLU outputs and the gradients through them rounded to bf16
real_conv2d, real_lrelu = F.conv2d, F.leaky_relu
rq = lambda v: v.to(torch.bfloat16).float()


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return rq(x)

    @staticmethod
    def backward(ctx, g):
        return rq(g)


def run_variant(wq, aq):
    F.conv2d = (lambda x, w, *a, **k: real_conv2d(x, w + (rq(w) - w).detach(), *a, **k)) if wq else real_conv2d
    F.leaky_relu = (lambda x, *a, **k: _RoundGrad.apply(real_lrelu(x, *a, **k))) if aq else real_lrelu
    try:
        return hybrid_grads(calls)[1]
    finally:
        F.conv2d, F.leaky_relu = real_conv2d, real_lrelu


for vn, (wq, aq) in {"weights bf16": (True, False), "activations + gradients bf16": (False, True), "both": (True, True)}.items():
    hv = run_variant(wq, aq)
    for task in ("m", "s"):
        cs = [cos(hv[k], hybrid[k]) for k in keys if k.startswith(task + ".") and float(hybrid[k].norm()) > 1e-3]
        cm = [cos(hv[k], mine[k]) for k in keys if k.startswith(task + ".") and k in mine and float(hybrid[k].norm()) > 1e-3]
        print("%-30s D.%s: cos(variant, fp32 hybrid) median %.4f min %.4f   cos(variant, mine) median %.4f" % (
            vn, task, sorted(cs)[len(cs) // 2], min(cs), sorted(cm)[len(cm) // 2]))
# cancellation between the two domain calls: |g_r + g_s| / |g_r|
for task in ("m", "s"):
    one = [c for c in calls if c[2] == (task == "m")]
    _, g_r = hybrid_grads([one[0]])
    k = task + ".Advent.2.module.weight_bar"
    print("D.%s layer 2: |g_r| %.4g   |g_r + g_s| %.4g" % (task, float(g_r[k].norm()), float(hybrid[k].norm())))
