"""The built library's device code, disassembled: no op_sel-modified packed-fp32 instruction outside the development kernel
that exists to demonstrate them.

Round 3 (R5 DESIGN 4.6): on gfx950 ``v_pk_mul_f32`` / ``v_pk_add_f32`` with op_sel modifiers (one half of a source broadcast,
halves crossed) changed their results whenever another stream ran MFMA / LDS-DMA kernels on the same chip
(tools/check_pk_opsel_concurrent.py); plain packed operations on fully defined register pairs and scalar operations did not.
The compiler forms such instructions on its own (SLP / loop vectoriser), so the build switches both vectorisers off and this
test keeps it that way.  Runs on the CPU: llvm-objdump only."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
OBJDUMP = Path("/opt/rocm/lib/llvm/bin/llvm-objdump")


@pytest.mark.parametrize("libname", ["libcgan_hip.so", "libcgan_hip_dev.so"])
def test_no_op_sel_modified_packed_fp32_instructions(tmp_path, libname):
    lib = ROOT / "climategan_amd" / libname
    if not lib.exists() or not OBJDUMP.exists():
        pytest.skip("needs the built library and llvm-objdump")
    shutil.copy(lib, tmp_path / lib.name)                       # --offloading writes the code objects next to its input
    subprocess.run([str(OBJDUMP), "--offloading", lib.name], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(tmp_path.glob("*gfx950*"))
    assert objs, "no gfx950 code object in the library"
    bad, kernels_seen = [], 0
    for o in objs:
        text = subprocess.run([str(OBJDUMP), "-d", str(o)], check=True, capture_output=True, text=True).stdout
        cur = ""
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                kernels_seen += 1
                continue
            if re.search(r"\bv_pk_(mul|add|fma)_f32\b.*op_sel", line) and "pk_opsel_kernel" not in cur:
                bad.append((cur, line.strip()))
    assert kernels_seen > 100
    assert not bad, "op_sel-modified packed-fp32 instructions in: %s" % sorted({b[0] for b in bad})[:5]
