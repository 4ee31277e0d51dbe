#!/usr/bin/env python3
"""Build-time check of the hidden prefetch ring (pandora_amd/csrc/pmx_buf.h, hring_load / hring_take): in every kernel of an object
file that loads into the ring's registers, NOTHING but the ring's own two statements may touch a VGPR from the ring's first register
on - the loads `buffer_load_dword* v[ring], v<low>, s[..]` and the takes `v_mov_b32 v<low>, v<ring>`.  The compiler does not know the ring's
registers are live between a load and its take; it allocates from v0 upwards and these kernels need 50-110 registers, so it never
gets there - this script turns "never" into a failed build.  Usage: check_hring.py <object file> <first ring register> [<name filter>]
(the filter is a regular expression on the mangled kernel name)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
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
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
    return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)


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
            if offenders or not takes:
                bad.append((head.group(1), loads, takes, offenders[:5]))
    return users, bad


def main():
    obj, first = sys.argv[1], int(sys.argv[2])
    users, bad = check(obj, first, sys.argv[3] if len(sys.argv) > 3 else "")
    for name, loads, takes, offenders in bad:
        print(f"check_hring: {obj}: {name}: {loads} ring loads, {takes} takes; other instructions touch v{first}..:", offenders)
    print(f"check_hring: {obj}: {users} kernels use the ring from v{first}, {len(bad)} violations")
    sys.exit(1 if bad or not users else 0)


if __name__ == "__main__":
    main()
