"""One-off (GPU box): the stale-packed-weight regression test must FAIL when ops.touch is disabled."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import climategan_amd.ops as o
import climategan_amd.optim as op
o.touch = lambda *a: None
op.touch = o.touch
rc = pytest.main(["tests/test_gpu_train.py", "-q", "-x", "-k", "forward_uses"])
print("EXPECTED-FAIL" if rc != 0 else "UNEXPECTED-PASS")
