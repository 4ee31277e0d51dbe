#!/usr/bin/env python3
"""Instruction histogram of the kernels of a gfx950 .s file (hipcc -S --cuda-device-only): whole function and its longest basic
block run (the unrolled main loop).  Usage: python tools/isa_hist.py file.s <substring of the mangled kernel name> ..."""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    for pat in sys.argv[2:]:
        for m in re.finditer(r"^(_Z\w*" + re.escape(pat) + r"\w*):.*\n", s, re.M):
            name = m.group(1)
            end = s.index(".Lfunc_end", m.end())
            body = s[m.end():end]
            lines = [ln.strip() for ln in body.splitlines()]
            # basic blocks
            blocks, cur = [], []
            for ln in lines:
                if not ln or ln.startswith((";", ".")) and not ln.startswith(".LBB"):
                    continue
                if ln.startswith(".LBB"):
                    blocks.append(cur)
                    cur = []
                    continue
                cur.append(ln.split()[0])
                if ln.split()[0].startswith(("s_cbranch", "s_branch")):
                    blocks.append(cur)
                    cur = []
            blocks.append(cur)
            allops = collections.Counter(op for b in blocks for op in b)
            big = max(blocks, key=len)
            vg = re.search(re.escape(name) + r"\n(?:.*\n){0,40}?\s+\.vgpr_count:\s+(\d+)", s)
            sg = re.search(r"\.sgpr_count:\s+(\d+)\n(?:.*\n){0,12}?\s+\.symbol:\s+" + re.escape(name), s)
            print(f"== {name}  total {sum(allops.values())}  vgpr {vg.group(1) if vg else '?'}")
            print(f"   largest block: {len(big)} instructions")
            print("   ", collections.Counter(big).most_common(24))


if __name__ == "__main__":
    main()
