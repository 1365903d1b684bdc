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


def test_no_register_is_touched_while_an_untracked_load_into_it_is_in_flight():
    """The Winograd kernels issue LDS reads from inline asm with hand-written s_waitcnt: the
    compiler takes such an asm's output for valid at once and may copy the register or hand it to
    something else before the data arrives.  (It did, with asm global loads: memory faults when
    two processes shared the GPU.)  tools/asm_hazard_check.py walks every path of the
    disassembly with the vmcnt / lgkmcnt queues and reports any such touch."""
    chk = os.path.join(ROOT, "tools", "asm_hazard_check.py")
    checked = 0
    # (conv_c32.o: the only hand-issued loads of its Winograd kernel are LDS reads; following the
    # compiler-tracked vmcnt queue through its uniform branches too takes minutes)
    for obj, opts in (("conv_wino.o", []), ("conv_wgrad.o", []), ("conv_c32.o", ["--lds-only"])):
        path = os.path.join(BUILD, obj)
        if not os.path.exists(path):
            pytest.skip("no built objects (run __graft_entry__.build() first)")
        r = subprocess.run(["python3", chk] + opts + [path, "wino"], capture_output=True, text=True)
        m = re.search(r"(\d+) kernels checked, (\d+) hazards, (\d+) incomplete", r.stdout)
        assert m, r.stdout + r.stderr
        assert int(m.group(2)) == 0 and int(m.group(3)) == 0 and r.returncode == 0, r.stdout[-4000:]
        checked += int(m.group(1))
    assert checked >= 12, checked


def test_hazard_checker_sees_a_touched_in_flight_register():
    """The checker on a hand-made listing: a global load whose destination is overwritten by
    VALU address arithmetic before the wait (the fault's signature), and the clean version."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("asm_hazard_check", os.path.join(ROOT, "tools", "asm_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = """
	global_load_dwordx4 v[8:11], v1, s[2:3]   // 000000001000: 00000000
	v_or3_b32 v9, v2, v3, s4                   // 000000001008: 00000000
	s_waitcnt vmcnt(0)                         // 000000001010: 00000000
	v_add_f32_e32 v0, v8, v9                   // 000000001014: 00000000
	s_endpgm                                   // 000000001018: 00000000
""".splitlines()
    good = [l for l in bad if "v_or3" not in l]
    assert len(mod.check_kernel("k", bad)) == 1
    assert mod.check_kernel("k", good) == []
    # the same across a loop's back edge: the load of one iteration, the touch in the next
    loop = """
	v_mov_b32_e32 v9, 0                        // 000000001000: 00000000
	s_waitcnt vmcnt(1)                         // 000000001004: 00000000
	global_load_dwordx4 v[8:11], v1, s[2:3]   // 000000001008: 00000000
	s_cbranch_scc1 65531                       // 000000001010: 00000000
	s_waitcnt vmcnt(0)                         // 000000001014: 00000000
	s_endpgm                                   // 000000001018: 00000000
""".splitlines()
    hz = mod.check_kernel("k", loop)
    assert len(hz) == 1 and "v_mov_b32" in hz[0], hz
