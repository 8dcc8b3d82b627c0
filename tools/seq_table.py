"""Per-launch table of ONE launch sequence from a rocprofv3 kernel trace of a serial bench run (tools/kstats.sh keeps the
trace when KEEP_TRACE=1): kernel, grid, duration -- the octave of a launch can be read off its grid.
usage: seq_table.py <dir> [substring filter]"""
import csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
def name(r): return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
starts = [i for i, r in enumerate(rows) if name(r).startswith("k_reset_state")]
k = len(starts) - 2
seg = rows[starts[k]:starts[k + 1]]
tot = 0
for r in seg:
    n = name(r)
    if flt and flt not in n: continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    print(f"{dur:9.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size', '')):>9s}  {n[:70]}")
print(f"total {tot / 1e3:.3f} ms over {len(seg)} launches (sequence {k} of {len(starts)})")
