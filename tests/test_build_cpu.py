"""CPU (-m "not gpu"): properties of the built gfx950 code objects that need no GPU.

No kernel may spill: a hand-scheduled MFMA kernel that touches scratch memory silently loses
its register blocking (three kernels carried 36-44 B/lane of scratch unnoticed)."""
import glob
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "unet-implementations_amd", "csrc", "build")
REGS = os.path.join(ROOT, "tools", "kernel_regs.sh")


def _kernels():
    objs = sorted(glob.glob(os.path.join(BUILD, "*.o")))
    if not objs:
        pytest.skip("no built objects (run __graft_entry__.build() first)")
    out = []
    for o in objs:
        txt = subprocess.run(["bash", REGS, o], capture_output=True, text=True, check=True).stdout
        for line in txt.splitlines():
            m = re.match(r"(\S+)\s+vgpr (\d+) agpr (\d+) sgpr (\d+) lds (\d+) scratch (\d+)", line)
            if m:
                out.append((os.path.basename(o), m.group(1), int(m.group(2)), int(m.group(6))))
    return out


def test_no_kernel_uses_scratch_memory():
    ks = _kernels()
    assert len(ks) > 150, f"only {len(ks)} kernels parsed"
    spilling = [(o, k[:100], v, s) for o, k, v, s in ks if s > 0]
    assert not spilling, "kernels with scratch (register spills): " + repr(spilling)
