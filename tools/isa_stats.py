#!/usr/bin/env python3
"""Static instruction mix of the basic blocks of a gfx950 kernel, from `hipcc -S --cuda-device-only` output.

    python tools/isa_stats.py file.s <kernel-substring> [min_instructions]

Prints the kernel's register budget and, for every basic block with at least `min_instructions` instructions
(default 150: the unrolled row loops of the streaming kernels), the counts of VALU / DPP / SALU / VMEM / LDS /
waitcnt instructions -- the numbers DESIGN.md quotes as "instructions per row" come from here."""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
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
        while i < len(lines) and not lines[i].startswith("\t.end_amdhsa_kernel") and not re.match(r"^\s*s_endpgm", lines[i]):
            ln = lines[i].strip()
            lm = re.match(r"^(\.LBB\w+):", ln)
            if lm:
                blocks.append((label, cur))
                cur, label = [], lm.group(1)
            elif ln and not ln.startswith((";", ".", "//")):
                cur.append(ln.split()[0])
            i += 1
        blocks.append((label, cur))
        info = {}
        for j in range(i, min(i + 400, len(lines))):
            for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
                mm = re.match(r"^; %s: (\d+)" % key, lines[j])
                if mm and key not in info:
                    info[key] = int(mm.group(1))
        print(f"{name}\n    {info}")
        for label, ins in blocks:
            if len(ins) < minn:
                continue
            c = dict(valu=0, pk=0, dpp=0, salu=0, vmem=0, lds=0, wait=0, other=0)
            for op in ins:
                if op.startswith("v_"):
                    c["valu"] += 1
                    if op.endswith("_dpp"):
                        c["dpp"] += 1
                    if op.startswith("v_pk_"):
                        c["pk"] += 1              # packed f32: two lane-ops, issues at half rate (valu_bench)
                elif op.startswith("s_waitcnt"):
                    c["wait"] += 1
                elif op.startswith("s_"):
                    c["salu"] += 1
                elif op.startswith(("global_", "buffer_", "flat_")):
                    c["vmem"] += 1
                elif op.startswith("ds_"):
                    c["lds"] += 1
                else:
                    c["other"] += 1
            print(f"    {label:14s} n={len(ins):5d}  " + "  ".join(f"{k}={v}" for k, v in c.items()) + f"  valu_slots={c['valu'] + c['pk']}")
        i += 1


if __name__ == "__main__":
    main()
