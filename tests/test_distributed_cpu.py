"""N > 1 path of bench.py on CPU: world_size 2, gloo.  The launch contract: every rank runs K timed steps between
barriers, rank-specific synthetic shards differ, the whole-job time is the MAX over ranks and the JSON line aggregates
all ranks (per-domain sample slots of the joint train step: GLOBAL batch 32 per domain, 32 / N per rank -- strong scaling,
SURVEY 8d M1).  The data-path collective of the training step
(bucketed gradient all-reduce) is covered by the reducer tests below."""
import json
import os
import socket
import sys
import time
from pathlib import Path

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from climategan_amd import fill

    calls = {"n": 0}

    def step():
        calls["n"] += 1
        time.sleep(0.01 * (rank + 1))  # rank 1 is the slow one

    elapsed = bench.timed_steps(step, steps=5, warmup=2, barrier=dist.barrier)
    total = bench.max_over_ranks(elapsed, dist, torch.device("cpu"))
    # the headline's shard: samples [first, first + per) of the job's 32 per domain, each drawn from its own seed
    from climategan_amd.parallel import shard_range
    bench.H = bench.W = 16
    first, per = shard_range(bench.GLOBAL_BS, world, rank)
    shard = bench.joint_batch(per, rank, torch.device("cpu"), first=first)
    whole = bench.joint_batch(bench.GLOBAL_BS, 0, torch.device("cpu"), first=0)      # what a 1-GPU job trains on
    same = all(torch.equal(shard[d]["data"][k], whole[d]["data"][k][first:first + per])
               for d in shard for k in shard[d]["data"])
    line = bench.result_line(world, 5, 2, total, "bf16", per) if rank == 0 else None
    q.put((rank, calls["n"], elapsed, total, (first, per, same, float(shard["r"]["data"]["x"].sum())),
           json.dumps(line) if line else None))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_timing_and_aggregation():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, n0, e0, t0, s0, line), (r1, n1, e1, t1, s1, _) = res
    assert (n0, n1) == (7, 7)                      # 2 warm-up + 5 timed on every rank
    assert abs(t0 - t1) < 1e-9 and t0 >= max(e0, e1) - 1e-9   # whole-job time = slowest rank, same on all ranks
    assert e1 >= 5 * 0.02 * 0.9                    # the slow rank really ran its 5 timed steps
    # strong scaling: the two ranks hold the two halves of the SAME 32 samples per domain a one-GPU job trains on
    assert s0[:3] == (0, 16, True) and s1[:3] == (16, 16, True) and s0[3] != s1[3]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 32
    assert d["config"]["batch_per_domain_per_gpu"] == 16
    assert abs(d["value"] - 32 * 5 / t0) < 1e-2 and abs(d["raw_images_per_s"] - 3 * d["value"]) < 1e-2
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in d


def test_shard_range_is_the_32_over_n_split():
    """bench.py's N = 1 / 2 / 4 / 8 points: contiguous equal shares that tile the global batch; anything the all-reduce
    average would weigh wrongly is refused."""
    import pytest

    sys.path.insert(0, str(ROOT))
    from climategan_amd.parallel import shard_range
    for world in (1, 2, 4, 8, 16, 32):
        parts = [shard_range(32, world, r) for r in range(world)]
        assert all(n == 32 // world for _f, n in parts)
        assert [f for f, _n in parts] == list(range(0, 32, 32 // world))
    assert shard_range(32, 8, 7) == (28, 4)
    for bad in ((32, 3, 0), (32, 64, 0), (32, 8, 8), (32, 8, -1), (32, 0, 0)):
        with pytest.raises(ValueError):
            shard_range(*bad)


def _reducer_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climategan_amd.parallel import GradBucketReducer, broadcast_parameters
        torch.manual_seed(100 + rank)                                   # replicas start different ...
        net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5))
        frozen = torch.nn.Parameter(torch.randn(4), requires_grad=False)  # like spectral norm's u / v
        net.register_parameter("frozen", frozen)
        broadcast_parameters(net)                                       # ... and are made identical
        ref = [p.detach().clone() for p in net.parameters()]
        red = GradBucketReducer(net.parameters(), bucket_mb=0.001)      # tiny cap -> several buckets
        nb = len(red.buckets)
        outs = []
        for step in range(2):                                           # two steps: buckets re-arm
            net.zero_grad(set_to_none=True)
            x = torch.full((3, 7), float(rank + 1 + step))
            net(x).sum().backward()
            red.finish()
            outs.append([p.grad.clone() for p in net.parameters() if p.requires_grad])
        q.put((rank, nb, [t.numpy() for t in ref], [[g.numpy() for g in o] for o in outs]))
    finally:
        dist.destroy_process_group()


def test_grad_bucket_reducer_averages_over_ranks_gloo():
    """world_size 2 on gloo: after broadcast both ranks hold rank 0's parameters; after backward + finish() both hold
    the MEAN of the two ranks' gradients (checked against a single-process evaluation of both inputs)."""
    import multiprocessing as mp
    import numpy as np
    import torch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + 7
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, nb0, ref0, outs0), (_, nb1, ref1, outs1) = res
    assert nb0 == nb1 and nb0 >= 2
    for a, b in zip(ref0, ref1):
        assert np.array_equal(a, b)                                     # broadcast made replicas identical
    # expected: mean over ranks of the local gradients, from a local re-evaluation with rank 0's parameters
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 5))
    net.register_parameter("frozen", torch.nn.Parameter(torch.zeros(4), requires_grad=False))
    with torch.no_grad():
        for p, v in zip(net.parameters(), ref0):
            p.copy_(torch.from_numpy(v))
    for step in range(2):
        exp = None
        for rank in range(2):
            net.zero_grad(set_to_none=True)
            net(torch.full((3, 7), float(rank + 1 + step))).sum().backward()
            g = [p.grad.clone() for p in net.parameters() if p.requires_grad]
            exp = g if exp is None else [a + b for a, b in zip(exp, g)]
        exp = [e / 2 for e in exp]
        for o in (outs0[step], outs1[step]):
            for got, e in zip(o, exp):
                assert np.allclose(got, e.numpy(), rtol=1e-6, atol=1e-7)


def _order_worker(rank, world, port, q, grad_dtype_name):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climategan_amd.parallel import GradBucketReducer, broadcast_parameters
        torch.manual_seed(7)
        layers = torch.nn.ModuleList([torch.nn.Linear(6, 6) for _ in range(4)])
        broadcast_parameters(layers)
        gd = getattr(torch, grad_dtype_name)
        red = GradBucketReducer(layers.parameters(), bucket_mb=1.0, grad_dtype=gd)      # ONE bucket holds all 8 tensors
        assert len(red.buckets) == 1
        x = torch.full((2, 6), 0.25 * (rank + 1))
        outs, launched_early = [], []
        for step in range(3):
            layers.zero_grad(set_to_none=True)
            if step == 1:
                # parallel branches, layer 0's node created LAST -> its gradient arrives FIRST: the learned trigger (the
                # parameter that completed the bucket in step 0) fires while the other gradients are still None
                loss = sum(layers[i](x).sum() for i in (3, 2, 1, 0))
            else:
                h = x
                for l in layers:
                    h = torch.tanh(l(h))
                loss = h.sum()
            loss.backward()
            launched_early.append(red.buckets[0].work is not None)      # launched from a hook, before finish()?
            red.finish()
            outs.append([p.grad.clone().numpy() for p in layers.parameters()])
        q.put((rank, launched_early, red._learning, outs))
    finally:
        dist.destroy_process_group()


def _run_order(grad_dtype_name):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_order_worker, args=(r, 2, port, q, grad_dtype_name)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def _expected_order_grads():
    import torch
    torch.manual_seed(7)
    layers = torch.nn.ModuleList([torch.nn.Linear(6, 6) for _ in range(4)])
    exp = []
    for step in range(3):
        acc = None
        for rank in range(2):
            layers.zero_grad(set_to_none=True)
            x = torch.full((2, 6), 0.25 * (rank + 1))
            if step == 1:
                loss = sum(layers[i](x).sum() for i in (3, 2, 1, 0))
            else:
                h = x
                for l in layers:
                    h = torch.tanh(l(h))
                loss = h.sum()
            loss.backward()
            g = [p.grad.clone() for p in layers.parameters()]
            acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        exp.append([(a / 2).numpy() for a in acc])
    return exp


def test_reducer_survives_a_changed_gradient_order_gloo():
    """The learned-trigger scheme (hooks only on the parameter that completed each bucket in the first step) when the
    gradients of step 2 become ready in a DIFFERENT order: the trigger fires early, finds gradients missing, leaves the
    bucket to ``finish()``; the averages are still exact, and step 3 (original order) overlaps again."""
    import numpy as np
    res = _run_order("float32")
    exp = _expected_order_grads()
    for rank, launched_early, learning, outs in res:
        assert learning is False                                        # trigger hooks were learned in step 0
        assert launched_early == [True, False, True], launched_early   # step 1 fell back to finish(), step 2 overlapped
        for step in range(3):
            for got, e in zip(outs[step], exp[step]):
                assert np.allclose(got, e, rtol=1e-6, atol=1e-7), (rank, step)


def test_reducer_bf16_wire_format_gloo():
    """bf16 buckets (opt-in on RCCL: CGAN_DDP_BF16_GRADS=1; forced here on gloo): averages within bf16 rounding of the exact ones, gradients
    stay fp32 tensors, both ranks end up with identical values."""
    import numpy as np
    res = _run_order("bfloat16")
    exp = _expected_order_grads()
    for step in range(3):
        for a, b, e in zip(res[0][3][step], res[1][3][step], exp[step]):
            assert a.dtype == np.float32 and np.array_equal(a, b)
            assert np.abs(a - e).max() <= 2.0 ** -7 * np.abs(e).max() + 1e-12


def _accum_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climategan_amd.parallel import GradBucketReducer, broadcast_parameters
        torch.manual_seed(3)
        net = torch.nn.Sequential(torch.nn.Linear(5, 9), torch.nn.Tanh(), torch.nn.Linear(9, 2))
        broadcast_parameters(net)
        red = GradBucketReducer(net.parameters(), bucket_mb=0.0001)      # one bucket per tensor or so
        outs = []
        for step in range(3):                                           # step 0 learns the triggers, 1 and 2 use them
            net.zero_grad(set_to_none=True)
            xa = torch.full((2, 5), 0.5 * (rank + 1) + step)
            xb = torch.full((2, 5), -0.25 * (rank + 2) - step)
            net(xa).sum().backward()                                    # buckets fill and go on the wire ...
            net(xb).pow(2).sum().backward()                             # ... then every gradient is accumulated AGAIN
            red.finish()
            outs.append([p.grad.clone().numpy() for p in net.parameters()])
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def test_reducer_two_backward_calls_per_update_gloo():
    """Gradient accumulation (two ``backward()`` calls before one ``finish()``): the buckets launched during the first
    backward are stale once the second one has accumulated into them; ``finish()`` must exchange the FINAL gradients
    (round-2 review: the hooks assumed exactly one backward per update)."""
    import multiprocessing as mp
    import numpy as np
    import torch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(5, 9), torch.nn.Tanh(), torch.nn.Linear(9, 2))
    for step in range(3):
        acc = None
        for rank in range(2):
            net.zero_grad(set_to_none=True)
            net(torch.full((2, 5), 0.5 * (rank + 1) + step)).sum().backward()
            net(torch.full((2, 5), -0.25 * (rank + 2) - step)).pow(2).sum().backward()
            g = [p.grad.clone() for p in net.parameters()]
            acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        for rank in range(2):
            for got, e in zip(res[rank][1][step], acc):
                assert np.allclose(got, (e / 2).numpy(), rtol=1e-6, atol=1e-7), (rank, step)


def _subgraph_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climategan_amd.parallel import GradBucketReducer, broadcast_parameters
        torch.manual_seed(4)
        # ONE bucket holding two independent heads: head a's gradients arrive last in the joint backward (the learned
        # trigger is one of its parameters), head b alone is reached by the second backward
        a, b = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
        net = torch.nn.ModuleList([a, b])
        broadcast_parameters(net)
        red = GradBucketReducer(net.parameters(), bucket_mb=25.0)
        outs = []
        for step in range(3):                                           # step 0 learns the trigger, 1 and 2 rely on it
            net.zero_grad(set_to_none=True)
            x = torch.full((2, 4), 0.5 * (rank + 1) + step)
            (a(x).sum() + b(x).pow(2).sum()).backward()                 # the bucket goes on the wire ...
            b(-x).pow(2).sum().backward()                               # ... then ONLY head b accumulates again
            red.finish()
            outs.append([p.grad.clone().numpy() for p in net.parameters()])
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def test_reducer_second_backward_over_another_subgraph_gloo():
    """A second ``backward()`` that accumulates into a launched bucket without reaching its trigger parameter (a
    masker-only pass after a joint one) fires no hook once only the triggers keep theirs; ``finish()`` must still see that
    what went on the wire is stale (round-3 advisor finding) and exchange the final gradients."""
    import multiprocessing as mp
    import numpy as np
    import torch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgraph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(4)
    a, b = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    params = list(a.parameters()) + list(b.parameters())
    for step in range(3):
        want = [torch.zeros_like(p) for p in params]
        for rank in range(2):
            for p in params:
                p.grad = None
            x = torch.full((2, 4), 0.5 * (rank + 1) + step)
            (a(x).sum() + b(x).pow(2).sum()).backward()
            b(-x).pow(2).sum().backward()
            for w, p in zip(want, params):
                w += p.grad / 2
        for rank in range(2):
            for got, w in zip(res[rank][1][step], want):
                np.testing.assert_allclose(got, w.numpy(), rtol=1e-5, atol=1e-6)


def _one_rank_edit_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from climategan_amd.parallel import GradBucketReducer, broadcast_parameters
        torch.manual_seed(5)
        net = torch.nn.Sequential(torch.nn.Linear(4, 6), torch.nn.Tanh(), torch.nn.Linear(6, 3))
        broadcast_parameters(net)
        red = GradBucketReducer(net.parameters(), bucket_mb=0.0001)
        outs = []
        for step in range(3):
            net.zero_grad(set_to_none=True)
            net(torch.full((2, 4), 0.3 * (rank + 1) + step)).sum().backward()
            if rank == 1 and step >= 1:
                # against the contract, and on ONE rank only: an in-place edit of an exchanged gradient before finish()
                net[2].weight.grad.mul_(2.0)
            red.finish()                     # must not hang: both ranks re-exchange that bucket, because one of them has to
            outs.append(([p.grad.clone().numpy() for p in net.parameters()],
                         [p.grad.data_ptr() == v.data_ptr() for b in red.buckets for p, v in zip(b.params, b.views)]))
        q.put((rank, outs))
    finally:
        dist.destroy_process_group()


def test_reducer_agrees_on_re_exchanges_across_ranks_gloo():
    """Round 6: ``finish()`` decides per bucket whether it must be exchanged again from what THIS rank saw; the flags go through
    a MAX all-reduce first, so a reason only one rank has (here: rank 1 scales one gradient in place after its hook sent it)
    makes every rank issue the same collectives -- no hang -- and the result is the mean of what the ranks hold at
    ``finish()``.  Also: with the fp32 wire the gradients come back AS views of the buckets' flat buffers (no copy back)."""
    import multiprocessing as mp
    import numpy as np
    import torch
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_one_rank_edit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(4, 6), torch.nn.Tanh(), torch.nn.Linear(6, 3))
    for step in range(3):
        acc = None
        for rank in range(2):
            net.zero_grad(set_to_none=True)
            net(torch.full((2, 4), 0.3 * (rank + 1) + step)).sum().backward()
            if rank == 1 and step >= 1:
                net[2].weight.grad.mul_(2.0)
            g = [p.grad.clone() for p in net.parameters()]
            acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        for rank in range(2):
            grads, are_views = res[rank][1][step]
            assert all(are_views)
            for got, e in zip(grads, acc):
                assert np.allclose(got, (e / 2).numpy(), rtol=1e-6, atol=1e-7), (rank, step)
        for a, b in zip(res[0][1][step][0], res[1][1][step][0]):
            assert np.array_equal(a, b)
