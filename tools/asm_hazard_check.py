#!/usr/bin/env python3
"""Check the gfx950 kernels of a HIP object / shared library for registers that are touched
while a load into them may still be in flight.

The Winograd kernels issue LDS reads (and used to issue global loads) from inline asm, which the
compiler does not track: the s_waitcnt in front of the consumer is written by hand, and nothing
but this check keeps the register allocator from copying such a register - or handing it to
something else - before the data has arrived.  (That is not a theoretical concern: untracked
coefficient loads whose values crossed the loop's back edge faulted when two processes shared the
GPU and the loads took longer than the rest of the iteration.)

The check walks the disassembly of every kernel in program order, follows each backward branch
once (so a loop body is seen with the state its previous iteration leaves behind), keeps the
in-order queues behind vmcnt and lgkmcnt, and reports every instruction that reads or writes a
VGPR that a queued load has yet to deliver.

usage: tools/asm_hazard_check.py [--max-states=N] [--lds-only] <obj-or-so> [kernel-name-substring ...]
exit code 1 when a hazard is found.
"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(path):
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "dev.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        tg = subprocess.check_output([LLVM + "/clang-offload-bundler", "--list", "--type=o", "--input=" + fat], text=True)
        tgt = [l for l in tg.split() if "gfx950" in l][0]
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=" + tgt, "--output=" + co])
        return subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True)


REG = re.compile(r"\b([va])(?:(\d+)|\[(\d+):(\d+)\])")


def regs_of(operand_text):
    out = set()
    for m in REG.finditer(operand_text):
        kind = m.group(1)
        if m.group(2) is not None:
            out.add((kind, int(m.group(2))))
        else:
            for r in range(int(m.group(3)), int(m.group(4)) + 1):
                out.add((kind, r))
    return out


def split_kernels(text):
    kernels, name, body = {}, None, []
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:\s*$", line)
        if m and not m.group(1).startswith("L"):
            if name:
                kernels[name] = body
            name, body = m.group(1), []
            continue
        if name is not None:
            body.append(line)
    if name:
        kernels[name] = body
    return kernels


def parse(body):
    """-> list of (address, mnemonic, operands); llvm-objdump prints `insn  // ADDRESS: ENCODING`"""
    out = []
    for line in body:
        m = re.search(r"//\s*([0-9A-Fa-f]+):", line)
        text = line.split("//")[0].strip()
        if not text or not m or text.startswith("."):
            continue
        parts = text.split(None, 1)
        out.append((int(m.group(1), 16), parts[0], parts[1] if len(parts) > 1 else ""))
    return out


def is_vm(mn):
    return mn.startswith(("buffer_", "global_", "flat_", "scratch_", "image_"))


def is_lds(mn):
    return mn.startswith("ds_")


def apply_wait(ops, vmq, lgq):
    vm = lg = None
    m = re.search(r"vmcnt\((\d+)\)", ops)
    if m:
        vm = int(m.group(1))
    m = re.search(r"lgkmcnt\((\d+)\)", ops)
    if m:
        lg = int(m.group(1))
    if not re.search(r"cnt\(", ops):      # raw immediate form (gfx9 layout)
        m = re.match(r"\s*(0x[0-9a-f]+|\d+)", ops)
        if m:
            v = int(m.group(1), 0)
            vm = (v & 0xF) | ((v >> 14) & 0x3) << 4
            lg = (v >> 8) & 0xF
    if vm is not None:
        vmq = vmq[max(0, len(vmq) - vm):]
    # scalar loads return out of order: only lgkmcnt(0) says anything with one in flight
    if lg is not None and (lg == 0 or None not in lgq):
        lgq = lgq[max(0, len(lgq) - lg):]
    return vmq, lgq


def canon(q, cap):
    """Operations without a register to deliver matter only as `younger` ones: drop them from the
    old end; the hardware counter saturates at `cap`."""
    k = 0
    while k < len(q) and q[k] is not None and not q[k]:
        k += 1
    q = q[k:]
    return q[-cap:] if len(q) > cap else q


def check_kernel(name, body, max_states=20000, track_vm=True):
    """Walk every path of the kernel (both sides of each conditional branch; a (branch target,
    queue state) pair is expanded once).  The queues hold, per outstanding operation, the frozen
    set of VGPRs it will write (empty for stores / DMA; None for a scalar load)."""
    ins = parse(body)
    index_of = {a: i for i, (a, _, _) in enumerate(ins)}
    hazards = {}
    seen = set()
    stack = [(0, (), ())]
    states = 0
    while stack and states < max_states:
        i, vmq, lgq = stack.pop()
        states += 1
        while i < len(ins):
            addr, mn, ops = ins[i]
            if mn == "s_waitcnt":
                vmq, lgq = apply_wait(ops, vmq, lgq)
                i += 1
                continue
            touched = regs_of(ops)
            first = regs_of(ops.split(",")[0])
            vm, lds = is_vm(mn), is_lds(mn)
            vm_load = vm and "load" in mn and "lds" not in mn
            lds_load = lds and re.match(r"ds_(read|load|bpermute|permute|swizzle|.*_rtn)", mn)
            for q, what in ((vmq, "vmcnt"), (lgq, "lgkmcnt")):
                # a load's own destination may be the destination of an older load of the same
                # queue: they return in order
                t = touched
                if (what == "vmcnt" and vm_load) or (what == "lgkmcnt" and lds_load):
                    t = regs_of(",".join(ops.split(",")[1:]))
                for age, d in enumerate(q):
                    if d and t & d:
                        key = (addr, what)
                        hazards.setdefault(key, "%s @%x: `%s %s` touches %s while a %s load into it is in flight (%d younger)"
                                           % (name, addr, mn, ops.strip(), ["%s%d" % r for r in sorted(t & d)[:4]],
                                              what, len(q) - 1 - age))
            if vm and track_vm:
                dst = frozenset(first) if vm_load or ("atomic" in mn and re.search(r"\b(glc|sc0)\b", ops)) else frozenset()
                vmq = vmq + (dst,)
            elif lds:
                lgq = lgq + (frozenset(first) if lds_load else frozenset(),)
            elif mn.startswith(("s_load", "s_buffer_load")):
                lgq = lgq + (None,)
            if mn == "s_endpgm":
                break
            if mn.startswith(("s_cbranch", "s_branch")):
                m = re.match(r"\s*(-?\d+)", ops)      # simm16, in dwords from the next instruction
                off = int(m.group(1))
                off = off - 65536 if off >= 32768 else off
                tgt = index_of.get(addr + 4 + 4 * off)
                if tgt is not None:
                    vmq, lgq = canon(vmq, 63), canon(lgq, 15)
                    key = (tgt, vmq, lgq)
                    if key not in seen:
                        seen.add(key)
                        stack.append((tgt, vmq, lgq))
                if mn == "s_branch":
                    break
            i += 1
    if states >= max_states:
        hazards[("limit", "")] = "INCOMPLETE %s: state limit reached" % name
    return list(hazards.values())


def main():
    args = sys.argv[1:]
    limit, track_vm = 20000, True
    while args and args[0].startswith("--"):
        a = args.pop(0)
        if a.startswith("--max-states="):
            limit = int(a.split("=")[1])
        elif a == "--lds-only":     # kernels whose only hand-issued loads are LDS reads: the
            track_vm = False        # vmcnt queue (all compiler-tracked) multiplies the states
        else:
            raise SystemExit("unknown option " + a)
    path, pats = args[0], args[1:]
    kernels = split_kernels(disassemble(path))
    bad = []
    n = 0
    for name, body in kernels.items():
        if pats and not any(p in name for p in pats):
            continue
        n += 1
        bad += check_kernel(name, body, limit, track_vm)
    seen = set()
    for h in bad:
        if h not in seen:
            print(h)
            seen.add(h)
    inc = sum(h.startswith("INCOMPLETE") for h in seen)
    print("%d kernels checked, %d hazards, %d incomplete" % (n, len(seen) - inc, inc))
    return 1 if seen else 0


if __name__ == "__main__":
    sys.exit(main())
