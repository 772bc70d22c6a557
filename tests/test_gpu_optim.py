"""GPU parity of the fused multi-tensor ExtraAdam (HIP) against the golden trajectory of the reference's ExtraAdam
(climategan/optim.py:200-291): extrapolation / step / extrapolation / step on three tensors.  fp32 elementwise
arithmetic; tolerance 2e-6 relative (the CPU reference and the kernel contract multiply-adds differently)."""
import numpy as np
import pytest
import torch

from helpers import golden_cases, load_golden, t
from oracle.make_golden import case_inputs

pytestmark = pytest.mark.gpu


def test_extra_adam_matches_reference_golden():
    from climategan_amd.optim import ExtraAdam

    case = golden_cases()["extra_adam"]
    gold = load_golden("extra_adam")
    inp = case_inputs("extra_adam", case)
    n = len(case["shapes"])
    params = [torch.nn.Parameter(t(inp["p%d" % i]).cuda()) for i in range(n)]
    opt = ExtraAdam(params, lr=case["lr"], betas=tuple(case["betas"]))
    with pytest.raises(RuntimeError, match="extrapolation before"):
        opt.step()
    for st in range(case["steps"]):
        for i, p in enumerate(params):
            p.grad = t(inp["g%d_%d" % (i, st)]).cuda()
        (opt.extrapolation if st % 2 == 0 else opt.step)()
        for i, p in enumerate(params):
            ref = gold["p%d_after%d" % (i, st)]
            got = p.data.cpu().numpy()
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()), (i, st)
    for i, p in enumerate(params):
        for key, gk in (("exp_avg", "m%d" % i), ("exp_avg_sq", "v%d" % i)):
            got, ref = opt.state[p][key].cpu().numpy(), gold[gk]
            assert np.abs(got - ref).max() <= 2e-6 * np.abs(ref).max(), key
        assert opt.state[p]["step"] == case["steps"]


def test_extra_adam_skips_params_without_grad_and_large_tensor():
    from climategan_amd.optim import ExtraAdam
    from oracle import cpu_ref

    g = torch.Generator().manual_seed(0)
    ps = [torch.randn(300_001, generator=g), torch.randn(17, generator=g)]
    params = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    opt = ExtraAdam(params, lr=2e-5, betas=(0.5, 0.999), weight_decay=0.01)
    ref = cpu_ref.ExtraAdamRef([p.clone() for p in ps], lr=2e-5, betas=(0.5, 0.999), weight_decay=0.01)
    for st in range(2):
        g0 = torch.randn(300_001, generator=g)
        params[0].grad = g0.cuda()
        params[1].grad = None                      # e.g. spectral-norm u/v (requires_grad=False)
        (opt.extrapolation if st == 0 else opt.step)()
        (ref.extrapolation if st == 0 else ref.step)([g0, None])
    assert (params[0].data.cpu() - ref.params[0]).abs().max() < 1e-6
    assert torch.equal(params[1].data.cpu(), ps[1])
