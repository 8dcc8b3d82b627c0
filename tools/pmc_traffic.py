#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM traffic.

Usage (on the GPU box, two separate passes as MI355X_MICROARCH.md prescribes -- FETCH_SIZE and
WRITE_SIZE do not fit one pass):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline
    python tools/pmc_traffic.py gpurun_out/pmcF gpurun_out/pmcW profiles/r03_pmc_traffic gpurun_out/pmcF.log
(the 4th argument is the stdout of the FETCH_SIZE pass: pairs per launch sequence and the number of sequences come from its JSON line)

Units / corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide (16 B/lane) coalesced streaming read, so reads are doubled for
kernels whose loads are float4 streams (the FED kernel); WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import os
import glob
import json
import sys


def load(dirname, counter):
    tot = collections.defaultdict(float)
    n = collections.Counter()
    for path in glob.glob(dirname + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            tot[k] += float(r["Counter_Value"]) * 1024.0
            n[k] += 1
    return tot, n


def run_shape(log_path):
    """(pairs per launch sequence, launch sequences) of the profiled bench run, from the JSON line it printed"""
    try:
        for ln in open(log_path):
            if ln.startswith("{") and '"metric"' in ln:
                d = json.loads(ln)
                # every float-path launch sequence the run enqueued (warm-up, timed steps, the untimed gather pass, ...)
                return int(d["config"]["pairs_per_launch_sequence"]), int(d["config"].get("float_sequences_enqueued", int(d["steps"]) + int(d["warmup"])))
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    fdir, wdir, out = sys.argv[1:4]
    shape = run_shape(sys.argv[4]) if len(sys.argv) > 4 else None
    if shape is None:
        sys.exit("pmc_traffic.py needs the stdout log of the profiled bench run (4th argument): the pairs per launch sequence "
                 "and the number of sequences are read from its JSON line, not assumed")
    fetch, nf = load(fdir, "FETCH_SIZE")
    write, nw = load(wdir, "WRITE_SIZE")
    rows = []
    for k in sorted(fetch, key=lambda k: -fetch[k]):
        rows.append(dict(kernel=k, dispatches=nf[k], fetch_size_bytes=fetch[k], write_size_bytes=write.get(k, 0.0)))
    fed = [r for r in rows if "k_fed_multi" in r["kernel"] or "k_fed_sf" in r["kernel"]]    # the FED family
    launches = sum(r["dispatches"] for r in fed)
    fed_fetch = sum(r["fetch_size_bytes"] for r in fed)
    fed_write = sum(r["write_size_bytes"] for r in fed)
    summary = dict(
        note="FETCH_SIZE doubled (gfx950 reports half of a 16 B/lane streaming read); WRITE_SIZE as reported",
        fed_launches=launches,
        fed_hbm_bytes_per_launch=(2.0 * fed_fetch + fed_write) / max(1, launches),
        fed_fetch_size_bytes=fed_fetch, fed_write_size_bytes=fed_write, per_kernel=rows,
        pairs_per_launch_sequence=shape[0])
    # per kernel class (bench.py CLASS_KERNELS): bytes per launch sequence + the hash of the sources the pass ran on, so that
    # bench.py refuses the figure once a kernel of the class has changed.  FETCH_SIZE is doubled only for the classes whose
    # loads are 16 B/lane streams (the case MI355X_MICROARCH.md calibrates); 4-byte gathers (describe / orient, NMS,
    # matcher staging) are taken as reported -- tools/pmc_gather_calib.py measures that case.
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import CLASS_KERNELS, class_source_hash
    nseq = shape[1]                                               # launch sequences in the profiled run (warmup + steps)
    classes = {}
    for k, names in CLASS_KERNELS.items():
        rs = [r for r in rows if any(n in r["kernel"] for n in names)]
        f = sum(r["fetch_size_bytes"] for r in rs)
        w = sum(r["write_size_bytes"] for r in rs)
        dbl = k in ("fed", "hessian", "prologue")
        classes[k] = dict(fetch_size_bytes=f, write_size_bytes=w, fetch_doubled=dbl, dispatches=sum(r["dispatches"] for r in rs),
                          hbm_bytes_per_sequence=((2.0 if dbl else 1.0) * f + w) / nseq, source_sha=class_source_hash(k))
    summary["classes"] = classes
    summary["launch_sequences"] = nseq
    json.dump(summary, open(out + ".json", "w"), indent=1)
    with open(out + ".csv", "w") as f:
        f.write("kernel,dispatches,FETCH_SIZE_bytes,WRITE_SIZE_bytes\n")
        for r in rows:
            f.write(f"\"{r['kernel']}\",{r['dispatches']},{r['fetch_size_bytes']:.0f},{r['write_size_bytes']:.0f}\n")
    print(json.dumps({k: v for k, v in summary.items() if k != "per_kernel"}))


if __name__ == "__main__":
    main()
