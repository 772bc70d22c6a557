"""Run bench.py in this process after setting development knobs of the library (same-box A/B of a kernel choice).
usage: python tools/bench_with_knobs.py gemm_ws=1 -- --no-cpu-baseline --sub-steps 0 --steps 10"""
import ctypes
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from climategan_amd import _lib  # noqa: E402

args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
lib = _lib.load_dev()
for kv in args[:split]:
    k, v = kv.split("=")
    getattr(lib, "cgan_debug_set_" + k)(ctypes.c_int(int(v)))
sys.argv = [str(ROOT / "bench.py")] + args[split + 1:]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
