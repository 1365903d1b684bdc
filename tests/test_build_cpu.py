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
    # every object in FULL mode (vmcnt and lgkmcnt queues; LDS-DMA at barriers; M0 wait state):
    # round 3 walked conv_c32.o with --lds-only because the path-by-path walk of its BSTATS kernel
    # took minutes - the walk is now per basic block with a merged-path fallback (seconds)
    for obj in ("conv_wino.o", "conv_wgrad.o", "conv_c32.o"):
        path = os.path.join(BUILD, obj)
        if not os.path.exists(path):
            pytest.skip("no built objects (run __graft_entry__.build() first)")
        r = subprocess.run(["python3", chk, path], capture_output=True, text=True)
        m = re.search(r"(\d+) kernels checked, (\d+) hazards, (\d+) incomplete", r.stdout)
        assert m, r.stdout + r.stderr
        assert int(m.group(2)) == 0 and int(m.group(3)) == 0 and r.returncode == 0, r.stdout[-4000:]
        checked += int(m.group(1))
    assert checked >= 150, checked


def _checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("asm_hazard_check", os.path.join(ROOT, "tools", "asm_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_hazard_checker_sees_an_lds_dma_the_barrier_wait_does_not_cover():
    """The Winograd loops stage U by global -> LDS DMA and make it visible with a hand-counted
    `s_waitcnt vmcnt(N)` + s_barrier: N = the VMEM instructions issued BEHIND the DMA pieces.  A
    listing where one of those loads is missing (what a compiler change could do: merge, move or
    drop it) leaves the last DMA piece inside the N youngest - flagged; the correct count is clean;
    an LDS-DMA right behind the SALU write of M0 (no wait state) is flagged too."""
    mod = _checker()

    def listing(n_loads_after, wait, nop=True):
        lines = ["	s_mov_b32 m0, s4                          // 000000001000: 00000000"]
        if nop:
            lines.append("	s_nop 0                                    // 000000001004: 00000000")
        lines.append("	global_load_lds_dwordx4 v1, s[2:3]        // 000000001008: 00000000")
        for k in range(n_loads_after):
            lines.append("	buffer_load_dwordx4 v[%d:%d], v2, s[8:11], 0 offen // %012x: 00000000"
                         % (8 + 4 * k, 11 + 4 * k, 0x1010 + 8 * k))
        lines += ["	s_waitcnt vmcnt(%d)                      // 000000001040: 00000000" % wait,
                  "	s_barrier                                  // 000000001044: 00000000",
                  "	ds_read_b64 v[40:41], v3                  // 000000001048: 00000000",
                  "	s_waitcnt vmcnt(0) lgkmcnt(0)             // 00000000104c: 00000000",
                  "	s_endpgm                                   // 000000001050: 00000000"]
        return lines

    assert mod.check_kernel("k", listing(2, 2)) == []
    hz = mod.check_kernel("k", listing(1, 2))           # one load went missing: the DMA is uncovered
    assert len(hz) == 1 and "DMA" in hz[0] and "s_barrier" in hz[0], hz
    assert mod.check_kernel("k", listing(1, 1)) == []
    hz = mod.check_kernel("k", listing(2, 2, nop=False))
    assert len(hz) == 1 and "M0" in hz[0], hz
    # the same through the merged-path fallback (state budget exhausted at once)
    hz = mod.check_kernel("k", listing(1, 2), max_states=0)
    assert len(hz) == 1 and "merged-path" in hz[0], hz
    assert mod.check_kernel("k", listing(2, 2), max_states=0) == []
    # across a loop's back edge: the DMA of one iteration, the barrier at the top of the next
    loop = """
	s_waitcnt vmcnt(1)                         // 000000001000: 00000000
	s_barrier                                  // 000000001004: 00000000
	s_mov_b32 m0, s4                           // 000000001008: 00000000
	s_nop 0                                    // 00000000100c: 00000000
	global_load_lds_dwordx4 v1, s[2:3]         // 000000001010: 00000000
	s_cbranch_scc1 65529                       // 000000001018: 00000000
	s_waitcnt vmcnt(0)                         // 00000000101c: 00000000
	s_endpgm                                   // 000000001020: 00000000
""".splitlines()
    hz = mod.check_kernel("k", loop)
    assert len(hz) == 1 and "DMA" in hz[0], hz
    assert mod.check_kernel("k", [l.replace("vmcnt(1)", "vmcnt(0)") for l in loop]) == []


def test_hazard_checker_flags_a_miscounted_wait_in_the_real_winograd_kernel(tmp_path):
    """End to end on the product source: conv_wino.hip rebuilt with its loop-top wait one too
    lenient (vmcnt(5) where four VMEM instructions follow the DMA pieces) must be flagged."""
    csrc = os.path.join(ROOT, "unet-implementations_amd", "csrc")
    src = open(os.path.join(csrc, "conv_wino.hip")).read()
    assert "s_waitcnt vmcnt(4) lgkmcnt(0)" in src
    bad = tmp_path / "conv_wino_bad.hip"
    bad.write_text(src.replace("s_waitcnt vmcnt(4) lgkmcnt(0)", "s_waitcnt vmcnt(5) lgkmcnt(0)"))
    obj = tmp_path / "bad.o"
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                    "-fno-gpu-rdc", "-I", csrc, "-c", str(bad), "-o", str(obj)], check=True,
                   capture_output=True)
    r = subprocess.run(["python3", os.path.join(ROOT, "tools", "asm_hazard_check.py"), str(obj)],
                       capture_output=True, text=True)
    assert r.returncode == 1 and "DMA" in r.stdout and "conv_wino_kernel" in r.stdout, r.stdout[-2000:]


def test_hazard_checker_sees_a_touched_in_flight_register():
    """The checker on a hand-made listing: a global load whose destination is overwritten by
    VALU address arithmetic before the wait (the fault's signature), and the clean version."""
    mod = _checker()
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
