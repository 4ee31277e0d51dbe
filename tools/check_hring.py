#!/usr/bin/env python3
"""Build-time check of the hidden prefetch ring (pandora_amd/csrc/pmx_buf.h, hring_load / hring_take): in every kernel of an object
file that loads into the ring's registers, NOTHING but the ring's own two statements may touch a VGPR from the ring's first register
on - the loads `buffer_load_dword* v[ring], v<low>, s[..]` and the takes `v_mov_b32 v<low>, v<ring>`.  The compiler does not know the ring's
registers are live between a load and its take; it allocates from v0 upwards and these kernels need 50-110 registers, so it never
gets there - this script turns "never" into a failed build.

Second check (round 6): the ring's hand-written `s_waitcnt vmcnt(N)` is right only if exactly N memory instructions
(loads and stores count alike on gfx9, in order) are issued between a slot's load and the wait in front of its take.  The source
derives N from what it believes a step issues; a compiler that merges two stores, or an edit that makes one conditional, would make
the wait too short - stale costs, silently.  So the marching loop is walked as the hardware runs it (the loop body twice, then the
code behind it) and for every take the memory instructions since the load of ITS register are counted: fewer than N fails the build
(the wait would not cover the load), more than N fails it too (the wait would be longer than the ring needs - the stall DESIGN 7.27
removed).

Usage: check_hring.py <object file> <first ring register> [<name filter>] [--llvm <llvm bin dir>] [--arch gfx950]
(the filter is a regular expression on the mangled kernel name; the Makefile passes --llvm / --arch from $(HIPCC) / $(ARCH))"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ARCH = "gfx950"
HERE = os.path.dirname(os.path.abspath(__file__))


def vregs(text):
    out = []
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out += list(range(int(a), int(b) + 1))
    out += [int(x) for x in re.findall(r"\bv(\d+)\b", text)]
    return out


def disassemble(obj, tmp):
    fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co")
    subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                           f"--targets=hipv4-amdgcn-amd-amdhsa--{ARCH}", f"--output={co}"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", co], text=True)


VMEM = ("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic", "flat_load", "flat_store",
        "flat_atomic", "scratch_load", "scratch_store")


def parse(fn):
    """[(address, mnemonic, operands, text)] of one function of the disassembly (llvm-objdump -d prints `text // ADDRESS: words`)"""
    out = []
    for line in fn.splitlines()[1:]:
        m = re.match(r"\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if not m or not m.group(1):
            continue
        ins = m.group(1)
        mnem = ins.split()[0]
        ops = [o.strip() for o in ins.split(None, 1)[1].split(",")] if " " in ins else []
        out.append((int(m.group(2), 16), mnem, ops, ins))
    return out


def branch_target(addr, mnem, ops):
    """s_branch / s_cbranch_*: simm16 dwords relative to the next instruction"""
    if not (mnem == "s_branch" or mnem.startswith("s_cbranch_")) or not ops:
        return None
    try:
        imm = int(ops[0], 0)
    except ValueError:
        return None
    if imm >= 0x8000:
        imm -= 0x10000
    return addr + 4 + 4 * imm


def wait_counts(fn, first, last=255):
    """Every take of the marching loop against its wait: [(text, vmcnt N, memory instructions since the register's load)] that differ."""
    ins = parse(fn)
    is_ring = lambda r: first <= r <= last
    ring_load = lambda mnem, ops: mnem.startswith("buffer_load_dword") and ops and all(is_ring(r) for r in vregs(ops[0])) and vregs(ops[0])
    ring_take = lambda mnem, ops: mnem.startswith("v_mov_b32") and len(ops) == 2 and vregs(ops[1]) and all(is_ring(r) for r in vregs(ops[1]))
    index = {a: i for i, (a, _, _, _) in enumerate(ins)}
    # the marching loop: the smallest backward-branch range that holds ring loads AND ring takes
    best = None
    for i, (a, mnem, ops, _) in enumerate(ins):
        t = branch_target(a, mnem, ops)
        if t is None or t > a or t not in index:
            continue
        j = index[t]
        body = ins[j:i + 1]
        if any(ring_load(m, o) for _, m, o, _ in body) and any(ring_take(m, o) for _, m, o, _ in body):
            if best is None or i - j < best[1] - best[0]:
                best = (j, i)
    if best is None:
        return None, []
    j, i = best
    trace = ins[j:i + 1] + ins[j:i + 1] + ins[i + 1:]
    since = {}      # ring register -> memory instructions issued after its last load
    pending = None  # the last vmcnt seen and no memory instruction since
    group = []      # the takes behind that wait: (text, memory instructions since the register's load)
    wrong, takes = [], 0

    def close_group():
        # a slot is loaded by one or more instructions and taken behind ONE wait: the wait must cover the slot's youngest piece
        # exactly (N = the fewest memory instructions since any of its registers' loads), which covers the older pieces too
        nonlocal takes
        if group:
            takes += len(group)
            need = min(n for _, n in group)
            if pending_of_group is None or pending_of_group != need:
                wrong.append((group[0][0], pending_of_group, need))
            group.clear()

    pending_of_group = None
    for k, (a, mnem, ops, text) in enumerate(trace):
        if mnem == "s_endpgm":
            if k > 2 * (i - j + 1):
                break
            continue
        if mnem == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", text)
            if m:
                close_group()
                pending = int(m.group(1))
            continue
        if mnem.startswith(VMEM):
            close_group()
            for r in since:
                since[r] += 1
            if ring_load(mnem, ops):
                for r in vregs(ops[0]):
                    since[r] = 0
            pending = None
            continue
        if ring_take(mnem, ops) and k >= (i - j + 1):  # (the first walk of the body only sets the loop's state up)
            r = vregs(ops[1])[0]
            if r in since:
                if not group:
                    pending_of_group = pending
                group.append((text, since[r]))
    close_group()
    return takes, wrong


def check(obj, first, name_filter=""):
    last = 255
    with tempfile.TemporaryDirectory() as tmp:
        dis = disassemble(obj, tmp)
    bad, users = [], 0
    for fn in re.split(r"\n(?=[0-9a-f]+ <[^>]+>:\n)", dis):
        head = re.match(r"[0-9a-f]+ <([^>]+)>:", fn)
        if not head or not re.search(name_filter, head.group(1)):
            continue
        loads = takes = 0
        offenders = []
        for line in fn.splitlines()[1:]:
            ins = line.split("//")[0].strip()
            if not ins:
                continue
            regs = vregs(ins)
            if not regs or max(regs) < first:
                continue
            ops = [o.strip() for o in ins.split(None, 1)[1].split(",")] if " " in ins else []
            mnem = ins.split()[0]
            if mnem.startswith("buffer_load_dword") and ops and all(first <= r <= last for r in vregs(ops[0])) \
                    and all(r < first for o in ops[1:] for r in vregs(o)):
                loads += 1
            elif mnem.startswith("v_mov_b32") and len(ops) == 2 and all(r < first for r in vregs(ops[0])) \
                    and all(first <= r <= last for r in vregs(ops[1])):
                takes += 1
            else:
                offenders.append(ins)
        if loads:
            users += 1
            counted, wrong = wait_counts(fn, first, last)
            if counted is None:
                offenders.append("no marching loop with ring loads and takes found: the wait counts could not be checked")
            elif not counted:
                offenders.append("no take of the marching loop could be matched with its load")
            for text, n, have in wrong[:5]:
                offenders.append(f"`{text}` behind s_waitcnt vmcnt({n}): {have} memory instructions since its register's load")
            if offenders or not takes:
                bad.append((head.group(1), loads, takes, offenders[:5]))
    return users, bad


def main():
    global LLVM, ARCH
    argv = sys.argv[1:]
    for flag in ("--llvm", "--arch"):
        if flag in argv:
            k = argv.index(flag)
            if flag == "--llvm":
                LLVM = argv[k + 1]
            else:
                ARCH = argv[k + 1]
            del argv[k:k + 2]
    obj, first = argv[0], int(argv[1])
    users, bad = check(obj, first, argv[2] if len(argv) > 2 else "")
    for name, loads, takes, offenders in bad:
        print(f"check_hring: {obj}: {name}: {loads} ring loads, {takes} takes; offending:", offenders)
    print(f"check_hring: {obj}: {users} kernels use the ring from v{first}, {len(bad)} violations")
    sys.exit(1 if bad or not users else 0)


if __name__ == "__main__":
    main()
