"""Which Python lines of the joint train step (bench.py's headline) issue torch-side device ops -- fills, copies, casts,
adds: each is a launch of its own between the HIP kernels.  usage (GPU box): python tools/trace_torch_ops_train.py"""
import collections
import sys
import traceback
from pathlib import Path

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

device = torch.device("cuda", 0)
T = bench.build_trainer(device, torch.bfloat16)
batch = bench.joint_batch(bench.SLICE_BS, 0, device)
T.G.painter.set_latent_shape((bench.SLICE_BS, 3, bench.H, bench.W), True)
counts, nbytes = collections.Counter(), collections.Counter()
SKIP = ("view", "_unsafe_view", "detach", "alias", "slice", "select", "expand", "as_strided", "unsqueeze", "squeeze", "t",
        "transpose", "permute", "reshape", "empty", "empty_like", "empty_strided", "unbind", "split", "_local_scalar_dense",
        "is_pinned", "lift_fresh", "is_contiguous", "size", "stride", "numel", "dim", "sym_size", "sym_numel",
        "sym_stride", "sym_storage_offset", "storage_offset", "is_same_size", "new_empty", "_pin_memory", "set_")


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in SKIP:
            st = [f for f in traceback.extract_stack()[:-1] if "climategan_amd" in f.filename]
            site = "%s:%d" % (st[-1].filename.split("/")[-1], st[-1].lineno) if st else "autograd engine"
            if name == "add" and args and isinstance(args[0], torch.Tensor) and args[0].dim() == 4:   # the engine's fan-in sums
                site = "%s fan-in %s %s" % (site, tuple(args[0].shape), str(args[0].dtype).replace("torch.", ""))
            if site == "autograd engine" and args and isinstance(args[0], torch.Tensor):
                site = "autograd engine %s %s" % (tuple(args[0].shape), str(args[0].dtype).replace("torch.", ""))
            counts[(func.__name__, site)] += 1
            t = args[0] if args and isinstance(args[0], torch.Tensor) else None
            if t is not None:
                nbytes[(func.__name__, site)] += t.numel() * t.element_size()
        return func(*args, **(kwargs or {}))


for _ in range(2):
    T.train_step(batch)
torch.cuda.synchronize()
# (the engine runs a device's backward on its own thread, where a thread-local dispatch mode is not active: one thread here)
with torch.autograd.set_multithreading_enabled(False), Mode():
    T.train_step(batch)
torch.cuda.synchronize()
print("torch-side ops of one joint train step:", sum(counts.values()))
for (name, site), n in counts.most_common(90):
    print("%5d  %-32s %-60s %9.3f MB" % (n, name, site, nbytes[(name, site)] / 1e6))
