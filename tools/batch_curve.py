#!/usr/bin/env python3
"""pairs/s of the 1080p path against the number of pairs in flight (contexts x pairs per launch sequence): where does the
batched path cross 2 000 pairs/s?  Prints one JSON object.  GPU box only."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import akaze_hip as ah  # noqa: E402
from akaze_hip import synth  # noqa: E402


def main():
    w, h = 1920, 1080
    p = ah.iAlignUp(w, 128)
    prs = [synth.pair(w, h, 1 + i) for i in range(8)]
    rows = []
    combos = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4), (2, 4), (4, 2), (1, 8), (2, 8), (4, 4), (2, 16), (4, 8), (2, 32), (2, 64)]
    if len(sys.argv) > 1:
        combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for nctx, ppseq in combos:
        nimg = 2 * ppseq
        d = torch.from_numpy(np.stack([synth.to_float(prs[(i // 2) % 8][i % 2], p) for i in range(nimg)])).cuda()
        for serial in (True, False):
            pipe = bench.Pipeline(ah, w, h, p, nimg, 10000, nctx, serial=serial, torch_stream=False)
            steps = max(20, 400 // (nctx * ppseq))
            rate = bench.timed_throughput(pipe, d, ppseq, steps, 5)
            rows.append(dict(contexts=nctx, pairs_per_sequence=ppseq, pairs_in_flight=nctx * ppseq, one_stream_per_context=serial,
                             pairs_per_s=round(rate, 1), ms_per_sequence=round(1e3 * ppseq / rate * nctx, 3)))
            print(rows[-1], file=sys.stderr, flush=True)
            pipe.close()
        del d
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
