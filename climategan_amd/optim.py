"""Host-side mirror of the reference's ``climategan/optim.py`` optimizer: ExtraAdam (extragradient Adam).

Same constructor and the reference's two-phase protocol -- ``extrapolation()`` on even steps, ``step()`` on odd
steps (trainer.py:674-694) -- with the whole update of every parameter tensor fused into ONE HIP launch
(``cgan_extra_adam_multi_tensor``).  State layout follows torch.optim conventions (``state[p] = {step, exp_avg,
exp_avg_sq}``) so ``state_dict()`` round-trips with the reference's checkpoints (``g_opt`` / ``d_opt``,
trainer.py:403-420).
"""
import ctypes as C

import torch
from torch.optim import Optimizer

from . import _lib
from ._lib import AdamItem
from .ops import _ptr, _stream


class ExtraAdam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= eps:
            raise ValueError("Invalid epsilon value: {}".format(eps))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if amsgrad:
            raise NotImplementedError("ExtraAdam: amsgrad=True has no HIP path (the reference never enables it)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        self.params_copy = {}   # id(p) -> saved parameters (reference: list self.params_copy, optim.py:149)
        self._has_copy = False

    def _run(self, mode):
        lib = _lib.load()
        for group in self.param_groups:
            # parameters whose Adam step count differs (a grad was None at some point) go in separate launches
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    if mode == 0 and not self._has_copy:
                        self.params_copy[id(p)] = p.data.clone()   # reference saves every parameter (optim.py:166-168)
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.data.is_contiguous():
                    raise RuntimeError("ExtraAdam (HIP): contiguous fp32 device parameters expected")
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p.data)
                    st["exp_avg_sq"] = torch.zeros_like(p.data)
                st["step"] += 1
                if mode == 0 and not self._has_copy:
                    self.params_copy[id(p)] = torch.empty_like(p.data)
                by_step.setdefault(st["step"], []).append(p)
            beta1, beta2 = group["betas"]
            for step, ps in by_step.items():
                items = (AdamItem * len(ps))()
                mx = 0
                for i, p in enumerate(ps):
                    st = self.state[p]
                    g = p.grad.data.contiguous()
                    if mode == 1 and id(p) not in self.params_copy:
                        raise RuntimeError("Need to call extrapolation before calling step.")
                    items[i] = AdamItem(p.data.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                        st["exp_avg_sq"].data_ptr(), self.params_copy[id(p)].data_ptr(), p.numel())
                    mx = max(mx, p.numel())
                # pinned + non_blocking: a pageable copy would make the host wait here for the whole backward pass
                # instead of running ahead into the next forward; the pinned buffer is kept until the next call
                host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).pin_memory()
                table = host.to(ps[0].device, non_blocking=True)
                self._tables = getattr(self, "_tables", [])[-7:] + [(host, table)]
                _lib.check(lib.cgan_extra_adam_multi_tensor(_ptr(table), len(ps), mx, mode,
                                                            int(mode == 0 and not self._has_copy), step, group["lr"],
                                                            beta1, beta2, group["eps"], group["weight_decay"],
                                                            _stream()), "cgan_extra_adam_multi_tensor")

    @torch.no_grad()
    def extrapolation(self):
        """Extrapolation step; saves a copy of the current parameters on the first call (optim.py:153-172)."""
        self._run(0)
        self._has_copy = True

    @torch.no_grad()
    def step(self, closure=None):
        """Update step applied to the parameters saved by ``extrapolation`` (optim.py:174-197)."""
        if not self._has_copy:
            raise RuntimeError("Need to call extrapolation before calling step.")
        loss = closure() if closure is not None else None
        self._run(1)
        self.params_copy = {}
        self._has_copy = False
        return loss
