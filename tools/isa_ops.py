#!/usr/bin/env python3
"""Opcode histogram of the largest basic blocks of one kernel (from `hipcc -S --cuda-device-only` output):
    python tools/isa_ops.py file.s <kernel-substring> [blocks]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lines = open(path).read().splitlines()
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if not (m and pat in m.group(1)):
        i += 1
        continue
    name = m.group(1)
    blocks, cur, label = [], [], "entry"
    i += 1
    while i < len(lines) and not re.match(r"^\s*s_endpgm", lines[i]) and not lines[i].startswith("\t.end_amdhsa_kernel"):
        ln = lines[i].strip()
        lm = re.match(r"^(\.LBB\w+):", ln)
        if lm:
            blocks.append((label, cur))
            cur, label = [], lm.group(1)
        elif ln and not ln.startswith((";", ".", "//")):
            cur.append(ln.split()[0])
        i += 1
    blocks.append((label, cur))
    print(name)
    for label, ins in sorted(blocks, key=lambda b: -len(b[1]))[:nb]:
        c = collections.Counter(ins)
        print(f"  {label}: {len(ins)} instructions")
        print("    " + ", ".join(f"{k} {v}" for k, v in c.most_common(45)))
    break
