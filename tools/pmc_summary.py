#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected (csv output), and -- with --table -- the
derived per-kernel table quoted in DESIGN.md / committed under profiles/ (wave time split, instructions per wave).

    rocprofv3 --pmc A B C --output-format csv -d gpurun_out/pmcX -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmcX [kernel-substring]
    python tools/pmc_summary.py --table gpurun_out/sqA gpurun_out/sqB > profiles/r02_sq_counters.txt

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over all waves of a dispatch; WAIT_ANY (parked on
s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY are disjoint and add up to the wave's lifetime
(MI355X_MICROARCH.md, rocprofv3 PMC slots).  One VALU instruction keeps its wave "active" for ~1 quad-cycle, so
valu% of ONE wave x waves per SIMD x 1/2 approximates the SIMD's VALU-pipe occupancy (2-cycle issue).
"""
import collections
import csv
import glob
import sys


def load(dirs, filt=""):
    tot = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.defaultdict(collections.Counter)
    for d in dirs:
        for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                if filt not in k:
                    continue
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
                n[k][r["Counter_Name"]] += 1
    return tot, n


def table(dirs):
    tot, n = load(dirs)
    avg = {k: {c: tot[k][c] / n[k][c] for c in tot[k]} for k in tot}
    hdr = (f"{'kernel':44s} {'disp':>4s} {'wave Mqc':>9s} {'wait%':>6s} {'stall%':>6s} {'act%':>5s} {'valu%':>6s} {'lds%':>5s} "
           f"{'VALU/wave':>9s} {'LDS/wave':>8s} {'VMEM_RD':>8s} {'VMEM_WR':>8s} {'waves':>9s}")
    print(hdr)
    for k, v in sorted(avg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = v.get("SQ_WAVE_CYCLES", 0)
        if wc < 1e7:
            continue
        w = v.get("SQ_WAVES", 1) or 1
        g = lambda c: v.get(c, 0.0)
        print(f"{k[:44]:44s} {max(n[k].values()):4d} {wc / 1e6:9.0f} {100 * g('SQ_WAIT_ANY') / wc:6.1f} {100 * g('SQ_WAIT_INST_ANY') / wc:6.1f} "
              f"{100 * g('SQ_ACTIVE_INST_ANY') / wc:5.1f} {100 * g('SQ_ACTIVE_INST_VALU') / wc:6.1f} {100 * g('SQ_ACTIVE_INST_LDS') / wc:5.1f} "
              f"{g('SQ_INSTS_VALU') / w:9.0f} {g('SQ_INSTS_LDS') / w:8.0f} {g('SQ_INSTS_VMEM_RD') / w:8.1f} {g('SQ_INSTS_VMEM_WR') / w:8.1f} {w:9.0f}")


def cache_table(dirs):
    """TCP / TCC / TA counters per kernel (tools/refresh_profiles.sh): L1 accesses, L2 requests and hit rates, the share of the
    kernel during which the texture-address unit is held up by the L1.  GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles,
    TA_ADDR_STALLED_BY_TC_CYCLES_sum the 256 CUs' -- hence the 32."""
    tot, n = load(dirs)
    avg = {k: {c: tot[k][c] / n[k][c] for c in tot[k]} for k in tot}
    print(f"{'kernel':44s} {'disp':>4s} {'L1 acc M':>9s} {'L2 req M':>9s} {'L1 hit%':>7s} {'L2 hit%':>7s} {'L2 miss MB':>10s} {'TA stalled%':>11s} {'TLB miss k':>10s}")
    for k, v in sorted(avg.items(), key=lambda kv: -kv[1].get("TCP_TOTAL_CACHE_ACCESSES_sum", 0)):
        g = lambda c: v.get(c, 0.0)
        acc, req = g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCP_TCC_READ_REQ_sum")
        if acc < 1e6:
            continue
        hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
        gui = g("GRBM_GUI_ACTIVE")
        print(f"{k[:44]:44s} {max(n[k].values()):4d} {acc / 1e6:9.1f} {req / 1e6:9.1f} {100 * (1 - req / acc) if acc else 0:7.1f} "
              f"{100 * hit / (hit + miss) if hit + miss else 0:7.1f} {miss * 64 / 1e6:10.0f} "
              f"{100 * g('TA_ADDR_STALLED_BY_TC_CYCLES_sum') / 32 / gui if gui else 0:11.1f} {g('TCP_UTCL1_TRANSLATION_MISS_sum') / 1e3:10.1f}")


def main():
    if sys.argv[1] == "--table":
        table(sys.argv[2:])
        return
    if sys.argv[1] == "--cache":
        cache_table(sys.argv[2:])
        return
    d = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    tot, n = load([d], filt)
    for k in sorted(tot):
        names = sorted(tot[k])
        disp = max(n[k].values())
        print(f"{k}  dispatches={disp}")
        for c in names:
            print(f"    {c:28s} {tot[k][c] / n[k][c]:16.1f} /dispatch")


if __name__ == "__main__":
    main()
