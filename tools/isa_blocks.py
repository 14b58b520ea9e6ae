#!/usr/bin/env python3
"""Per basic block instruction mix of one kernel in a hipcc -S listing (tools for the instruction-diet work on the pre-filter).

usage: isa_blocks.py listing.s kernel-substring
"""
import re, sys
from collections import Counter

def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur, name = [], Counter(), "entry"
    order = []
    for l in lines[start + 1:end]:
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", s)
            if m:
                blocks.append((name, cur, order)); cur, name, order = Counter(), m.group(1), []
            continue
        op = s.split()[0]
        cls = ("mfma" if "mfma" in op else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
               "scratch" if op.startswith("scratch_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else
               "smem" if op.startswith("s_load") or op.startswith("s_buffer") else "wait" if op.startswith("s_waitcnt") else
               "branch" if op.startswith(("s_cbranch", "s_branch")) else "salu")
        cur[cls] += 1
        if cls == "branch":
            order.append(s.split()[-1])
            if op != "s_branch" or True:   # the fall-through after a branch is a block of its own
                blocks.append((name, cur, order)); cur, name, order = Counter(), name + "+", []
    blocks.append((name, cur, order))
    tot = Counter()
    for n, c, o in blocks:
        tot.update(c)
        print("%-12s valu %4d lds %3d mfma %2d vmem %3d scratch %3d salu %3d wait %3d  -> %s" % (
            n, c["valu"], c["lds"], c["mfma"], c["vmem"], c["scratch"], c["salu"] + c["smem"], c["wait"], ",".join(o)))
    print("static total", dict(tot))

if __name__ == "__main__":
    main()
