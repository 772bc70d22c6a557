"""GPU box: the packed-fp32 instruction pair of the first wildfire blur (cgan_debug_pk_opsel), alone and next to another
stream's wide-layer GEMM launches: do the results change?"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from climategan_amd import _lib, ops  # noqa: E402

lib = _lib.load_dev()
lib.cgan_debug_pk_opsel.restype = C.c_int
lib.cgan_debug_pk_opsel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
n = 1 << 22
g = torch.Generator(device="cuda"); g.manual_seed(0)
x = torch.rand(n, device="cuda", generator=g)
taps = torch.rand(512, device="cuda", generator=g)
dt = torch.float16
xg = ops.NHWC(torch.randn((16, 80, 80, 256), device="cuda", generator=g).to(dt), 256)
pwg = ops.pack_conv_weight(torch.randn(256, 256, 3, 3, device="cuda", generator=g) * 0.05, None, dt)
side = torch.cuda.Stream()
junk = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), dtype=torch.int32, device="cuda", generator=g)
for mode, name in ((0, "v_pk_mul/add_f32 with op_sel (the blur's pair)"), (1, "plain v_pk_mul/add_f32 on defined pairs"), (2, "scalar v_mul/v_add_f32")):
    outs = []
    for kind in ("alone", "next to conv_gemm", "next to conv_gemm", "alone"):
        out = torch.empty(2 * n, device="cuda")
        if kind != "alone":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(12):
                    ops.conv2d(xg, pwg, pad=1)
        _lib.check(lib.cgan_debug_pk_opsel(x.data_ptr(), junk.data_ptr(), taps.data_ptr(), out.data_ptr(), n, 64, mode, ops._stream()), "pk")
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        outs.append(out)
    print("%-48s values differing from the first (alone) run: %s" % (name, [int((o != outs[0]).sum()) for o in outs[1:]]), flush=True)
