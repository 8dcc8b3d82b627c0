"""Timeline of one single-image detectAndCompute call from a rocprofv3 kernel trace of hipakaze_demo.

usage: single_timeline.py <dir with *kernel_trace.csv> [call index]
Splits the trace at the state-reset kernel that opens every launch sequence, takes the call in the middle of the float-path
loop (or the given index), and prints per kernel: start offset, duration, queue -- then the busy / idle split of the call and the
per-kernel totals.  Writes <dir>/single_call.csv (the rows of that one call) for profiles/."""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def name(r):
    return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")


starts = [i for i, r in enumerate(rows) if name(r).startswith("k_reset_state")]
if not starts:
    sys.exit("no k_reset_state kernel in the trace")
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
seg = rows[starts[k]:starts[k + 1]]
t0 = int(seg[0]["Start_Timestamp"])
prev_end = int(rows[starts[k] - 1]["End_Timestamp"]) if starts[k] else t0
qcol = "Queue_Id" if "Queue_Id" in seg[0] else None
print(f"call {k} of {len(starts)}: {len(seg)} kernels, gap to the previous call's last kernel {(t0 - prev_end) / 1e3:.1f} us")
busy, cur_e = 0, t0
with open(d + "/single_call.csv", "w", newline="") as out:
    wr = csv.writer(out)
    wr.writerow(["kernel", "queue", "start_us", "dur_us", "grid", "workgroup"])
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = r[qcol] if qcol else ""
        g = r.get("Grid_Size_X", r.get("Grid_Size", ""))
        wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", ""))
        wr.writerow([name(r), q, f"{(s - t0) / 1e3:.2f}", f"{(e - s) / 1e3:.2f}", g, wg])
        print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} us  q{q:>3s}  grid {g:>8s}  {name(r)[:60]}")
        if e > cur_e:
            busy += e - max(s, cur_e)
            cur_e = e
wall = cur_e - t0
print(f"wall {wall / 1e3:.1f} us, some kernel running {busy / 1e3:.1f} us ({100.0 * busy / wall:.1f} %), "
      f"sum of kernel durations {sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3:.1f} us")
tot = {}
for r in seg:
    n = name(r)
    a = tot.setdefault(n, [0, 0])
    a[0] += 1
    a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, ns) in sorted(tot.items(), key=lambda t: -t[1][1])[:16]:
    print(f"  {n[:48]:48s} {c:3d} x  {ns / 1e3:8.1f} us")
