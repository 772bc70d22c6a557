"""Data-parallel gradient exchange for the training step: one process per GPU, RCCL through
``torch.distributed`` (backend "nccl" on ROCm), replicas of G and D, **bucketed all-reduce (average) of the gradients,
launched from post-accumulate-grad hooks while the rest of the backward is still running**.

Stock DistributedDataParallel cannot wrap ``OmniGenerator`` (it has no ``forward``; the trainer calls ``encode`` /
``paint`` / decoders directly), and this package's gradients come out of custom autograd Functions anyway, so the
reducer works on the parameter list:

* parameters are grouped into ~25 MB buckets in REVERSE registration order (the order gradients become ready: the
  Painter's last layers / D's output conv first);
* the wire format is **fp32** by default: the ranks' ``p.grad`` tensors are gathered into the bucket's flat buffer by one
  multi-tensor copy, summed by the collective and scaled by 1/world -- exactly the mean the single-process reference
  would see on the concatenated batch.  ``CGAN_DDP_BF16_GRADS=1`` (or ``grad_dtype=torch.bfloat16``) halves the bytes
  (210.8 MB instead of 421.7 MB per G exchange) at the price of one bf16 rounding of the inputs and of the sum
  (relative error <= 2^-8 per element, unbiased; tests/test_distributed_cpu.py) -- opt-in, because nothing in this
  repository can check it against an fp32 exchange on more than one RCCL rank; the chosen type is printed at setup;
* a bucket's all-reduce is issued (``async_op=True``, on RCCL's own stream) as soon as its last gradient has been
  accumulated; xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is per-link bound: few
  large messages, not one per tensor (105 M G parameters = 17 buckets);
* ``finish()`` waits for the outstanding buckets and hands the averaged values over as ``param.grad`` -- with the fp32 wire
  the gradients BECOME views of the bucket's flat buffer (no copy back) and the average is the collective's own (``AVG`` on
  RCCL; one in-place scale of the flat buffer elsewhere): one pass over the gradients per exchange (the gather) instead of
  three (round 5: gather, copy back, scale).  ExtraAdam needs it before both ``extrapolation()`` and ``step()`` (reference
  trainer.py:678-683);
* whether a bucket has to be exchanged AGAIN (a second ``backward()`` touched it after its hook sent it) is decided from
  rank-local observations, so ``finish()`` first agrees on it: the per-bucket flags go through one MAX all-reduce of a
  few bytes over a host-side (gloo) group -- no device synchronisation -- and every rank then issues the same collectives
  even if only one of them saw a reason (round 5 trusted every rank to see the same; a rank-conditional edit of a gradient
  would have left the ranks waiting on different collectives).

``CGAN_DDP_DIRECT_RCCL=1`` (RCCL backend only) takes torch out of the collective: the reducer creates its OWN communicator
through the C ABI (``cgan_rccl_load`` / ``cgan_comm_unique_id`` / ``cgan_comm_init_rank``: the unique id travels over the
existing process group) and enqueues ``cgan_allreduce_bucket`` on a stream of its own, ordered after the bucket's gather
by an event and before ``finish()``'s copy-back by another -- the path a host without torch.distributed would take
(INTEGRATION.md); the default stays ``dist.all_reduce``.

``broadcast_parameters`` makes replicas identical at start (parameters AND buffers, incl. the spectral-norm ``u``/``v``
vectors, which are parameters with ``requires_grad=False``).
"""
import os
from typing import List

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    """More than one rank in the default process group -- or, for the single-GPU test of the RCCL code path
    (tests/test_gpu_train.py), a one-rank group with CGAN_DDP_SINGLE_RANK_TEST=1."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CGAN_DDP_SINGLE_RANK_TEST") == "1"


def shard_range(global_batch: int, world: int, rank: int):
    """(first sample, count) of rank ``rank``'s share of a step's ``global_batch`` samples per domain: contiguous, equal
    shares (global batch 32 per domain -> 4 per GPU per domain on 8 GPUs; the reference's one process takes
    ``bs`` samples per domain per step, trainer.py:633,935-939).  Equal shares are REQUIRED, not a convenience: the
    gradient exchange averages the ranks' batch-mean gradients with equal weights, which is the global batch mean only
    then -- a batch the world size does not divide is refused."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("shard_range: rank %d outside a world of %d" % (rank, world))
    if global_batch < world or global_batch % world:
        raise ValueError("shard_range: a global batch of %d per domain cannot be split evenly over %d ranks (the all-reduce "
                         "average weighs the ranks equally)" % (global_batch, world))
    per = global_batch // world
    return rank * per, per


def broadcast_parameters(module: torch.nn.Module, src: int = 0) -> None:
    """Rank ``src``'s parameters and buffers to every rank (C3)."""
    if not is_distributed():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


class _EventWork:
    """``wait()`` of a collective enqueued on the reducer's own stream: the calling stream waits for its event."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class _Bucket:
    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat = None
        self.views = None
        self.trigger = None
        self.pending = len(params)
        self.work = None


class GradBucketReducer:
    """Bucketed, overlapped gradient averaging over the default process group."""

    def __init__(self, params, bucket_mb: float = 25.0, grad_dtype=None):
        self.params = [p for p in params if p.requires_grad]
        self.active = is_distributed()
        self.world = dist.get_world_size() if self.active else 1
        if grad_dtype is None:      # exact fp32 exchange unless bf16 is asked for (RCCL only: gloo has no bf16 sum)
            on_rccl = self.active and dist.get_backend() == "nccl"
            grad_dtype = torch.bfloat16 if (on_rccl and os.environ.get("CGAN_DDP_BF16_GRADS") == "1") else torch.float32
        self.grad_dtype = grad_dtype
        if os.environ.get("CGAN_DDP_BUCKET_MB"):          # (bucket-size experiments and the stream-ordering tests)
            bucket_mb = float(os.environ["CGAN_DDP_BUCKET_MB"])
        cap = int(bucket_mb * 2 ** 20)
        self.buckets: List[_Bucket] = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):                        # reverse registration ~ order of gradient readiness
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > cap:
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self.buckets.append(_Bucket(cur))
        self._bucket_of = {id(p): b for b in self.buckets for p in b.params}
        self.direct = False
        self._comm = None
        if self.active and dist.get_backend() == "nccl" and os.environ.get("CGAN_DDP_DIRECT_RCCL") == "1":
            self._init_direct()
        # streams on which gradients of these parameters are produced besides the one a hook happens to run on (the
        # trainer issues the Masker and the Painter branch of a backward on two streams): a bucket's gather waits for them
        self.streams = []
        if self.active and dist.get_rank() == 0:
            print("GradBucketReducer: %d parameters, %.1f MB in %d buckets, %s on the wire, %d ranks over %s%s"
                  % (len(self.params), sum(p.numel() for p in self.params) * torch.finfo(grad_dtype).bits / 8e6,
                     len(self.buckets), str(grad_dtype).split(".")[1], self.world, dist.get_backend(),
                     " (own communicator, cgan_allreduce_bucket)" if os.environ.get("CGAN_DDP_DIRECT_RCCL") == "1" else ""),
                  flush=True)
        # First step: a hook on EVERY parameter counts the bucket down and remembers which parameter completed it.
        # Afterwards only those trigger parameters keep a hook (the backward graph is the same every step): ~1500 calls
        # from the autograd engine into Python per step cost more (~15 ms of a 160 ms step) than the overlap buys.
        # host-side group for finish()'s per-bucket agreement (every rank constructs its reducers in the same order)
        self._flag_group = None
        if self.active and os.environ.get("CGAN_DDP_NO_FLAG_SYNC") != "1":
            # (also on the one-rank RCCL group of tests/test_gpu_train.py: the only place a single-GPU box can run a gloo group
            # beside an RCCL default group)
            try:
                self._flag_group = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
            except Exception as e:       # no gloo in this build, a refused rendezvous ...: rank-local decisions, as in round 5
                if dist.get_rank() == 0:
                    print("GradBucketReducer: no host-side group for the re-exchange flags (%s: %s); every rank decides for "
                          "itself" % (type(e).__name__, e), flush=True)
        self._learning = True
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]
        self.reset()

    def _init_direct(self):
        """Own RCCL communicator through the C ABI (include/climategan_hip.h, "Data-parallel gradient exchange")."""
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(lib.cgan_rccl_load(path.encode()), "cgan_rccl_load")
        uid = C.create_string_buffer(128)
        if dist.get_rank() == 0:
            _lib.check(lib.cgan_comm_unique_id(uid), "cgan_comm_unique_id")
        box = [uid.raw]
        dist.broadcast_object_list(box, src=0)
        uid = C.create_string_buffer(box[0], 128)
        comm = C.c_void_p()
        _lib.check(lib.cgan_comm_init_rank(C.byref(comm), self.world, uid, dist.get_rank()), "cgan_comm_init_rank")
        self._comm, self._lib = comm, lib
        self._comm_stream = torch.cuda.Stream()
        self._wire = {torch.float32: _lib.CGAN_F32, torch.bfloat16: _lib.CGAN_BF16, torch.float16: _lib.CGAN_F16}[self.grad_dtype]
        self.direct = True

    def reset(self):
        for b in self.buckets:
            b.pending = len(b.params)
            b.seen = set()
            b.stale = False
            b.work = None
            b.sent = None
            b.averaged = False

    def _views(self, b: _Bucket, grads):
        """The bucket's flat buffer and its per-parameter views (shaped like the gradients), allocated once."""
        if b.flat is None or b.flat.device != grads[0].device or b.flat.dtype != self.grad_dtype:
            b.flat = torch.empty(b.numel, dtype=self.grad_dtype, device=grads[0].device)
            b.views, off = [], 0
            for g in grads:
                b.views.append(b.flat[off:off + g.numel()].view(g.shape))
                off += g.numel()
        return b.views

    def _launch(self, b: _Bucket):
        grads = [p.grad for p in b.params]
        if self.streams and grads[0].is_cuda:
            cur = torch.cuda.current_stream(grads[0].device)
            for st in self.streams:
                if st != cur:
                    cur.wait_stream(st)
        # multi-tensor copies (a handful of launches per bucket instead of one per parameter: 1500 parameters)
        torch._foreach_copy_(self._views(b, grads), grads)
        # what went on the wire: finish() compares (a later backward() that accumulates into a gradient of this bucket
        # WITHOUT reaching its trigger parameter -- another sub-graph, e.g. a masker-only pass after a joint one -- fires
        # no hook once only the triggers keep theirs)
        b.sent = [(g.data_ptr(), g._version) for g in grads]
        b.averaged = False
        if self.direct:
            from . import _lib
            gathered = torch.cuda.Event()
            gathered.record()
            self._comm_stream.wait_event(gathered)
            _lib.check(self._lib.cgan_allreduce_bucket(b.flat.data_ptr(), b.numel, self._wire, self._comm,
                                                       self._comm_stream.cuda_stream), "cgan_allreduce_bucket")
            done = torch.cuda.Event()
            done.record(self._comm_stream)
            b.work = _EventWork(done)
            return
        # the mean over the ranks: RCCL averages inside the collective; gloo (and the direct path above) sum, finish() scales
        b.averaged = dist.get_backend() == "nccl" and self.grad_dtype == torch.float32
        b.work = dist.all_reduce(b.flat, op=dist.ReduceOp.AVG if b.averaged else dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        if not self.active:
            return
        b = self._bucket_of[id(p)]
        if b.work is not None:
            # a gradient of a bucket whose exchange is already in flight was accumulated again (a second backward()
            # before the optimizer step: gradient accumulation, two loss.backward() calls per update): what is on the
            # wire is stale -- finish() waits for it, discards it and exchanges the bucket again
            b.stale = True
            return
        if self._learning:
            b.seen.add(id(p))                     # per parameter, not per hook call: accumulations do not count twice
            b.pending = len(b.params) - len(b.seen)
            if b.pending == 0:
                b.trigger = p
                self._launch(b)
        elif all(q.grad is not None for q in b.params):
            # learned trigger: the bucket's last gradient of the first step.  Should the order ever differ, some gradient
            # is still None (zero_grad sets them to None) and the bucket is left to finish()
            b.pending = 0
            self._launch(b)

    def _keep_trigger_hooks_only(self):
        for h in self._hooks:
            h.remove()
        self._hooks = [b.trigger.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets
                       if getattr(b, "trigger", None) is not None]
        self._learning = False

    def finish(self):
        """Wait for every bucket (launching the ones whose parameters received no gradient this step as zeros would be
        wrong: parameters without a gradient are skipped on every rank alike) and write the averages back.

        Contract (advisor, round 4): between ``backward()`` and this call the gradients must not be touched in place (clipping,
        unscaling: do them AFTER ``finish()``).  A bucket whose gradients changed after its hook sent them (``sent`` records
        their data pointers and versions) is exchanged again here -- on EVERY rank as soon as one rank saw it (the flags'
        MAX all-reduce below), so the ranks always issue the same collectives.  After this call ``param.grad`` may be a view
        of the reducer's flat buffer (fp32 wire): valid until the next exchange of that bucket, i.e. through the optimizer
        step and ``zero_grad``."""
        if not self.active:
            self.reset()
            return
        # ---- what this rank would do per bucket: 0 nothing (no gradients at all), 1 wait for the exchange its hook started,
        # 2 exchange now (the hook never fired, or what it sent is stale) -- then the ranks agree on the maximum
        todo = []
        for b in self.buckets:
            if b.pending != 0:
                if all(p.grad is None for p in b.params):
                    todo.append(0)
                    continue
                missing = [p for p in b.params if p.grad is None]
                if missing:
                    raise RuntimeError("GradBucketReducer: %d parameters of a bucket got no gradient while others did; "
                                       "replicas would diverge" % len(missing))
                todo.append(2)
            else:
                grads = [p.grad for p in b.params]
                changed = b.stale or b.sent != [(g.data_ptr(), g._version) for g in grads]      # see _on_grad / _launch
                todo.append(2 if changed else 1)
        if self._flag_group is not None:
            flags = torch.tensor(todo, dtype=torch.uint8)
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self._flag_group)
            agreed = flags.tolist()
            for b, mine, theirs in zip(self.buckets, todo, agreed):
                if mine == 0 and theirs != 0:
                    raise RuntimeError("GradBucketReducer: another rank has gradients for a bucket this rank has none for; "
                                       "replicas would diverge")
            todo = agreed
        for b, what in zip(self.buckets, todo):
            if what == 0:
                continue
            if b.work is not None:
                b.work.wait()                     # (an exchange in flight is always completed before its buffer is reused)
            if what == 2:
                self._launch(b)
                b.work.wait()
            grads = [p.grad for p in b.params]
            if self.world > 1 and not b.averaged:
                b.flat.mul_(1.0 / self.world)                          # the average, one pass over the flat buffer
            if b.flat.dtype == grads[0].dtype:
                for p, v in zip(b.params, b.views):                     # the gradients ARE the bucket's views from here on
                    p.grad = v
            else:
                torch._foreach_copy_(grads, b.views)                   # 16-bit wire: back to the gradients' own dtype (fp32)
        if self._learning and any(getattr(b, "trigger", None) is not None for b in self.buckets):
            self._keep_trigger_hooks_only()
        self.reset()

    def remove(self):
        for h in self._hooks:
            h.remove()
        if self._comm is not None:
            torch.cuda.synchronize()
            self._lib.cgan_comm_destroy(self._comm)
            self._comm, self.direct = None, False
