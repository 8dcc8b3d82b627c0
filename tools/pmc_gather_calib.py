#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE for 4-byte gathers (the descriptor stage's access shape).

MI355X_MICROARCH.md calibrates the counter only for 16 B/lane streams (it reports half of their bytes on gfx950).  This
program launches libhipakaze's k_gather_probe -- every lane reads ONE dword from a pseudo-random 128-byte line of a buffer
far larger than the caches, a known number of times -- so that FETCH_SIZE divided by the number of gathers is what the
counter charges per gathered line:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/gcal -- python3 tools/pmc_gather_calib.py
    python3 tools/pmc_gather_calib.py --summarise gpurun_out/gcal

Reading: ~64 B per gather = one 64-byte request per line, counted at face value (no doubling for gathers); ~32 B would mean
the streaming-read halving applies to gathers too.
"""
import csv
import ctypes as C
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BYTES, BLOCKS, PER_LANE, ITERS = 4 << 30, 4096, 64, 4


def run():
    sys.path.insert(0, os.path.join(ROOT, "cuda-akaze_amd"))
    import akaze_hip as ah
    ms = C.c_double()
    ah.check(ah.lib.hak_op_gather_probe(BYTES, BLOCKS, PER_LANE, ITERS, C.byref(ms)))
    n = BLOCKS * 256 * PER_LANE
    print(f"gathers per launch {n}, {ms.value:.3f} ms per launch, {n * 64 / ms.value / 1e6:.1f} GB/s if every gather moves 64 B")


def summarise(d):
    tot, cnt = 0.0, 0
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_gather_probe" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                tot += float(r["Counter_Value"]) * 1024.0
                cnt += 1
    n = BLOCKS * 256 * PER_LANE
    print(f"k_gather_probe: {cnt} dispatches, FETCH_SIZE {tot / max(cnt, 1) / 1e6:.1f} MB per dispatch, "
          f"{tot / max(cnt, 1) / n:.1f} B per gathered dword ({n} gathers from distinct random 128-byte lines of a {BYTES >> 30} GiB buffer)")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        summarise(sys.argv[2])
    else:
        run()
