"""Host-side mirror of the reference's ``climategan/optim.py`` optimizer: ExtraAdam (extragradient Adam).

Same constructor and the reference's two-phase protocol -- ``extrapolation()`` on even steps, ``step()`` on odd
steps (trainer.py:674-694) -- with the whole update of every parameter tensor fused into ONE HIP launch
(``cgan_extra_adam_multi_tensor``).  State layout follows torch.optim conventions (``state[p] = {step, exp_avg,
exp_avg_sq}``) so ``state_dict()`` round-trips with the reference's checkpoints (``g_opt`` / ``d_opt``,
trainer.py:403-420).
"""
import torch
from torch.optim import Optimizer

from . import _lib
from ._lib import AdamItem
from .ops import _ptr, _stream, touch


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
                    # the reference saves a copy of EVERY parameter at the first extrapolation (optim.py:166-168); the
                    # copy of a gradient-less one is read back only if that parameter does receive a gradient by the
                    # following step() (domain batches that differ between the two calls).  Parameters that can never
                    # get one (requires_grad=False: the spectral-norm u / v vectors) are skipped: not observable.
                    if mode == 0 and not self._has_copy and p.requires_grad:
                        self.params_copy[id(p)] = p.data.clone()
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
                if _lib.CALL_LOG is not None:        # p, g, m, v read; p, m, v (+ the saved copy on an extrapolation) written
                    _lib.log_bytes(sum(p.numel() for p in ps) * 4 * (8 if mode == 0 else 7))
                _lib.check(lib.cgan_extra_adam_multi_tensor(_ptr(table), len(ps), mx, mode,
                                                            int(mode == 0 and not self._has_copy), step, group["lr"],
                                                            beta1, beta2, group["eps"], group["weight_decay"],
                                                            _stream()), "cgan_extra_adam_multi_tensor")
                # the kernel wrote the parameters through raw pointers: bump their version counters, or every cached
                # packed weight (norms._PackCache keys on them) stays the one packed before this update
                touch(*ps)

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


def get_scheduler(optimizer, hyperparameters, iterations=-1):
    """Learning-rate scheduler from ``<model>.opt`` (reference optim.py:10-51): ``constant`` / None -> no scheduler,
    ``step`` -> StepLR(lr_step_size, lr_gamma), ``multi_step`` -> MultiStepLR (``lr_milestones`` a list, or an int
    expanded to ``range(lr_milestones, 1000 or iterations, lr_step_size)``).  Host logic only (it edits
    ``param_groups[i]["lr"]``, which the HIP update reads per launch)."""
    from torch.optim import lr_scheduler

    get = hyperparameters.get if hasattr(hyperparameters, "get") else (lambda k: getattr(hyperparameters, k, None))
    policy, lr_step_size, lr_gamma, milestones = (get(k) for k in ("lr_policy", "lr_step_size", "lr_gamma",
                                                                   "lr_milestones"))
    if policy is None or policy == "constant":
        return None
    if policy == "step":
        return lr_scheduler.StepLR(optimizer, step_size=lr_step_size, gamma=lr_gamma, last_epoch=iterations)
    if policy == "multi_step":
        if isinstance(milestones, int):
            if lr_step_size is None:
                raise AssertionError("multi_step with an int lr_milestones needs lr_step_size")
            milestones = list(range(milestones, 1000 if iterations == -1 else iterations, lr_step_size))
        return lr_scheduler.MultiStepLR(optimizer, milestones=list(milestones), gamma=lr_gamma, last_epoch=iterations)
    # the reference RETURNS (does not raise) the exception object here (optim.py:48-50); callers would fail later on
    # ``scheduler.step()``.  Raising at once is the same failure, earlier.
    raise NotImplementedError("learning rate policy [%s] is not implemented" % policy)


def get_optimizer(net, opt_conf, tasks=None, is_disc=False, iterations=-1):
    """(optimizer, scheduler, lr_names) from ``opts.gen.opt`` / ``opts.dis.opt`` (reference optim.py:54-124): one
    parameter group over ``net.parameters()`` when ``lr`` is a float or holds only ``default``; otherwise one group per
    task with its own learning rate (G: encoder for "m", painter for "p", ``decoders[task]`` for the others; D:
    ``net[task]``).  Group and parameter order are the reference's, so ``state_dict()`` of the optimizer is
    interchangeable with the reference's ``g_opt`` / ``d_opt`` checkpoint entries.  Only ExtraAdam has a HIP update
    (the reference's default for both models, defaults.yaml:74,197); the other names raise."""
    lr_names = []
    lr = opt_conf.lr
    if tasks is None or isinstance(lr, float) or len(lr) == 1:
        lr_default = lr if isinstance(lr, float) else lr.default
        params = net.parameters()
        lr_names.append("full")
    else:
        lr_default = lr.default
        params = []
        for task in tasks:
            task_lr = lr.get(task, lr_default)
            parameters = None
            if not is_disc:
                if task == "m":
                    # the encoder rides on the masker's learning rate, as its own group ahead of decoders["m"]
                    params.append({"params": net.encoder.parameters(), "lr": task_lr})
                    lr_names.append("encoder")
                if task == "p":
                    if hasattr(net, "painter"):
                        parameters = net.painter.parameters()
                        lr_names.append("painter")
                else:
                    parameters = net.decoders[task].parameters()
                    lr_names.append("decoder_%s" % task)
            elif task in net:
                parameters = net[task].parameters()
                lr_names.append("disc_%s" % task)
            if parameters is not None:
                params.append({"params": parameters, "lr": task_lr})
    name = str(opt_conf.optimizer).lower()
    if name != "extraadam":
        raise NotImplementedError("get_optimizer: only ExtraAdam has a HIP update (got %r)" % opt_conf.optimizer)
    opt = ExtraAdam(params, lr=lr_default, betas=(opt_conf.beta1, 0.999))
    return opt, get_scheduler(opt, opt_conf, iterations), lr_names
