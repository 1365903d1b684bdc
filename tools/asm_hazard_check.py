#!/usr/bin/env python3
"""Check the gfx950 kernels of a HIP object / shared library for registers that are touched
while a load into them may still be in flight.

The Winograd kernels issue LDS reads (and used to issue global loads) from inline asm, which the
compiler does not track: the s_waitcnt in front of the consumer is written by hand, and nothing
but this check keeps the register allocator from copying such a register - or handing it to
something else - before the data has arrived.  (That is not a theoretical concern: untracked
coefficient loads whose values crossed the loop's back edge faulted when two processes shared the
GPU and the loads took longer than the rest of the iteration.)

The check walks the basic blocks of every kernel along every path (a block is expanded once per
distinct queue state, so a loop body is seen with what its previous iteration leaves in flight),
keeps the in-order queues behind vmcnt and lgkmcnt, and reports every instruction that reads or
writes a VGPR that a queued load has yet to deliver; a global -> LDS DMA still outstanding at an
s_barrier (the hand-counted `s_waitcnt vmcnt(N)` in front of it no longer covers the DMA); and an
LDS-DMA issued directly behind the SALU write of M0 it depends on (missing wait state).

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


# Queue entries are frozensets of tokens: the VGPRs / AGPRs a load will write, as ("v", n) /
# ("a", n) tuples, plus DMA for a global -> LDS DMA and SCALAR for a scalar load (both have no
# vector destination).  Stores are empty sets.
DMA, SCALAR = "lds-dma", "scalar"


def canon(q, cap):
    """Operations with nothing to deliver matter only as `younger` ones: drop them from the old
    end; the hardware counter saturates at `cap`."""
    k = 0
    while k < len(q) and not q[k]:
        k += 1
    q = q[k:]
    return q[-cap:] if len(q) > cap else q


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
    if lg is not None and (lg == 0 or not any(SCALAR in d for d in lgq)):
        lgq = lgq[max(0, len(lgq) - lg):]
    return vmq, lgq


def is_lds_dma(mn):
    """global_load_lds_* / buffer_load_* ... lds: VMEM loads whose destination is LDS (M0 base)."""
    return is_vm(mn) and "load" in mn and "lds" in mn


def branch_target(addr, ops, index_of):
    m = re.match(r"\s*(-?\d+)", ops)      # simm16, in dwords from the next instruction
    if not m:
        return None
    off = int(m.group(1))
    off = off - 65536 if off >= 32768 else off
    return index_of.get(addr + 4 + 4 * off)


def join(a, b):
    """May-be-in-flight union of two queue states, aligned at the YOUNG end (a counted wait keeps
    the N youngest operations)."""
    if len(a) < len(b):
        a, b = b, a
    pad = len(a) - len(b)
    return a[:pad] + tuple(x | y for x, y in zip(a[pad:], b))


class Walker:
    """Transfer function of one basic block + the hazards found on the way."""

    def __init__(self, name, body, track_vm):
        self.name, self.track_vm = name, track_vm
        self.ins = parse(body)
        self.index_of = {a: i for i, (a, _, _) in enumerate(self.ins)}
        self.leaders = {0}
        for i, (addr, mn, ops) in enumerate(self.ins):
            if mn.startswith(("s_cbranch", "s_branch")):
                tgt = branch_target(addr, ops, self.index_of)
                if tgt is not None:
                    self.leaders.add(tgt)
                self.leaders.add(i + 1)
            elif mn == "s_endpgm":
                self.leaders.add(i + 1)
        self.hazards = {}

    def block(self, i, vmq, lgq):
        """Run the block that starts at instruction i; returns [(successor, vmq, lgq), ...]."""
        ins, name = self.ins, self.name
        m0_fresh = False        # the previous instruction was an SALU write of M0
        while i < len(ins):
            addr, mn, ops = ins[i]
            if mn == "s_waitcnt":
                vmq, lgq = apply_wait(ops, vmq, lgq)
                m0_fresh = False
            else:
                touched = regs_of(ops)
                first = regs_of(ops.split(",")[0])
                vm, lds = is_vm(mn), is_lds(mn)
                dma = is_lds_dma(mn)
                vm_load = vm and "load" in mn and not dma
                lds_load = lds and re.match(r"ds_(read|load|bpermute|permute|swizzle|.*_rtn)", mn)
                for q, what in ((vmq, "vmcnt"), (lgq, "lgkmcnt")):
                    # a load's own destination may be the destination of an older load of the
                    # same queue: they return in order
                    t = touched
                    if (what == "vmcnt" and vm_load) or (what == "lgkmcnt" and lds_load):
                        t = regs_of(",".join(ops.split(",")[1:]))
                    for age, d in enumerate(q):
                        if t & d:
                            self.hazards.setdefault(
                                (addr, what),
                                "%s @%x: `%s %s` touches %s while a %s load into it is in flight (%d younger)"
                                % (name, addr, mn, ops.strip(), ["%s%d" % r for r in sorted(t & d)[:4]],
                                   what, len(q) - 1 - age))
                if dma and m0_fresh:
                    self.hazards.setdefault(
                        (addr, "m0"), "%s @%x: `%s %s` directly behind the SALU write of M0 (one wait "
                        "state required)" % (name, addr, mn, ops.strip()))
                if mn == "s_barrier" and any(DMA in d for d in vmq):
                    young = len(vmq) - 1 - max(k for k, d in enumerate(vmq) if DMA in d)
                    self.hazards.setdefault(
                        (addr, "dma"), "%s @%x: s_barrier with a global -> LDS DMA of this wave still in "
                        "flight (%d younger VMEM operations): the vmcnt wait in front of the "
                        "barrier does not cover it" % (name, addr, young))
                m0_fresh = mn.startswith("s_") and re.match(r"\s*m0\b", ops) is not None
                if vm and (self.track_vm or dma):
                    if dma:
                        dst = frozenset((DMA,))
                    elif vm_load or ("atomic" in mn and re.search(r"\b(glc|sc0)\b", ops)):
                        dst = frozenset(first)
                    else:
                        dst = frozenset()
                    vmq = vmq + (dst,)
                elif lds:
                    lgq = lgq + (frozenset(first) if lds_load else frozenset(),)
                elif mn.startswith(("s_load", "s_buffer_load")):
                    lgq = lgq + (frozenset((SCALAR,)),)
            if mn == "s_endpgm":
                return []
            if mn.startswith(("s_cbranch", "s_branch")):
                vmq, lgq = canon(vmq, 63), canon(lgq, 15)
                out = []
                tgt = branch_target(addr, ops, self.index_of)
                if tgt is not None:
                    out.append((tgt, vmq, lgq))
                if mn != "s_branch":
                    out.append((i + 1, vmq, lgq))
                return out
            i += 1
            if i in self.leaders:
                return [(i, canon(vmq, 63), canon(lgq, 15))]
        return []


def check_kernel(name, body, max_states=20000, track_vm=True):
    """Walk the kernel over its basic blocks with the in-order queues behind vmcnt / lgkmcnt.
    Reported:
      * an instruction that reads or writes a VGPR while a queued load has yet to deliver it;
      * a global -> LDS DMA of this wave still outstanding at an s_barrier: the kernels make DMA
        data visible to the other waves by `s_waitcnt vmcnt(N)` + barrier, with N counted by hand
        from the VMEM instructions issued behind the DMA - if the compiler moves, merges or
        drops one of those, the wait no longer covers the DMA and the other waves read stale LDS;
      * an LDS-DMA directly behind the SALU write of M0 it depends on (one wait state is
        required; nothing inserts it inside an asm block).
    First EXACTLY: every path, a (block, queue state) pair expanded once - a loop body is seen with
    what its previous iteration left in flight.  A kernel with many uniform branches around
    compiler-tracked loads (the BSTATS form of conv_wino32q_kernel) has more distinct states than
    that is worth: past `max_states` the walk is redone as a may-be-in-flight dataflow - ONE
    state per block, the union over all paths into it (aligned at the young end of the queues),
    iterated to its fixed point.  That is sound (nothing the exact walk would report is missed)
    and can only over-report."""
    w = Walker(name, body, track_vm)
    seen = set()
    stack = [(0, (), ())]
    states = 0
    complete = True
    while stack:
        key = stack.pop()
        if key in seen or key[0] >= len(w.ins):
            continue
        if states >= max_states:
            complete = False
            break
        seen.add(key)
        states += 1
        stack.extend(w.block(*key))
    if complete:
        return list(w.hazards.values())
    w = Walker(name, body, track_vm)
    state = {0: ((), ())}
    work = [0]
    rounds = 0
    while work:
        i = work.pop()
        rounds += 1
        if rounds > 200000:
            w.hazards[("limit", "")] = "INCOMPLETE %s: the merged walk did not converge" % name
            break
        for j, vmq, lgq in w.block(i, *state[i]):
            if j >= len(w.ins):
                continue
            old = state.get(j)
            new = (vmq, lgq) if old is None else (join(old[0], vmq), join(old[1], lgq))
            if new != old:
                state[j] = new
                work.append(j)
    return ["%s  [merged-path walk]" % h if not h.startswith("INCOMPLETE") else h
            for h in w.hazards.values()]


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
