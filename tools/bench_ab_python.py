"""Run bench.py in this process after setting module-level A/B switches of the Python side.
usage: python tools/bench_ab_python.py climategan_amd.deeplab.resnet101_v3.FUSE_RESIDUAL_GRADIENT=0 -- --steps 10 ..."""
import importlib
import runpy
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
for kv in args[:split]:
    k, v = kv.split("=")
    parts = k.split(".")
    for n in range(len(parts) - 1, 0, -1):           # longest importable module prefix, then attributes
        try:
            obj = importlib.import_module(".".join(parts[:n]))
            break
        except ImportError:
            continue
    for a in parts[n:-1]:
        obj = getattr(obj, a)
    setattr(obj, parts[-1], type(getattr(obj, parts[-1]))(int(v)))
sys.argv = [str(ROOT / "bench.py")] + args[split + 1:]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
