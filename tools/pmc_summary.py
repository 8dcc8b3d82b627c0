#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected (csv output).

    rocprofv3 --pmc A B C --output-format csv -d gpurun_out/pmcX -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmcX [kernel-substring]
"""
import collections
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            if filt not in k:
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
    for k in sorted(tot):
        names = sorted(tot[k])
        disp = max(n[k].values())
        print(f"{k}  dispatches={disp}")
        for c in names:
            print(f"    {c:28s} {tot[k][c] / n[k][c]:16.1f} /dispatch")


if __name__ == "__main__":
    main()
