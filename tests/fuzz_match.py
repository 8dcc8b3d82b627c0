"""Randomised parity run of the matchers alone: hak_match (cuMatch, akaze.cpp:55-64; with and without a context) and hak_match_knn2
on seeded random set sizes and descriptor populations, the matrix-core kernel and the VALU kernel, against the oracle.

Test infrastructure (loads the oracle as the checker).  `python tests/fuzz_match.py --cases 400 --seed 3` on the GPU box.
A case draws: n1, n2 from a mix of small / medium / large sizes with a third of them snapped to the kernels' tile geometry (32-row tiles,
128-query blocks, 192-row chunks, 1152-row slices, +-1); a descriptor population -- uniform random (no accepted matches), planted
near-copies (accepted matches), exact duplicates of train rows (ties between equal distances in different tiles / residue classes /
slices: the reference's accept rule, akazed.cu:2222-2236), a few all-zero / all-one rows; garbage in the struct padding (D9)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("cuda-akaze_amd", "oracle", ""):
    p = os.path.join(ROOT, sub)
    if p not in sys.path:
        sys.path.insert(0, p)

MFIELDS = ("match", "distance", "match_x", "match_y")


def draw_case(seed, index):
    rng = np.random.default_rng([seed, index, 77])
    pick = lambda seq: seq[int(rng.integers(len(seq)))]

    def size(allow_zero):
        kind = pick(("small", "medium", "medium", "large"))
        n = int({"small": rng.integers(0 if allow_zero else 1, 300), "medium": rng.integers(300, 3000), "large": rng.integers(3000, 12001)}[kind])
        if rng.random() < 0.33:
            n = max(0 if allow_zero else 1, pick((32, 64, 128, 192, 256, 384, 1152, 2304, 4096, 6912)) * int(rng.integers(1, 4)) + pick((-1, 0, 1)))
        return min(n, 12000)
    n1, n2 = max(1, size(False)), size(True)
    return dict(index=index, n1=n1, n2=n2, planted=pick((0.0, 0.3, 0.6)), dup=pick((0.0, 0.0, 0.2, 0.5)), maxflip=pick((10, 45, 120)),
                extremes=bool(rng.random() < 0.3), ctx=bool(rng.random() < 0.5), knn2=(pick(((1, 1), (4, 5), (2, 3))), bool(rng.random() < 0.5)),
                seed=int(rng.integers(1 << 30)))


def make_sets(ah, synth, c):
    rng = np.random.default_rng(c["seed"])
    n1, n2 = c["n1"], c["n2"]
    train = synth.random_descriptors(max(n2, 1), c["seed"] % 100003, ah.POINT_DTYPE)[:n2]
    if n2 > 4 and c["dup"] > 0:                                      # exact duplicates: equal distances in other tiles / classes / slices
        k = int(n2 * c["dup"])
        train["features"][rng.choice(n2, k, replace=False)] = train["features"][rng.integers(0, n2, k)]
    if n2 > 8 and c["extremes"]:
        train["features"][rng.integers(0, n2, 2)] = 0
        full = np.full(61, 0xFF, np.uint8)
        full[60] = 0x3F
        train["features"][rng.integers(0, n2, 2)] = full
    query = synth.random_descriptors(n1, (c["seed"] + 1) % 100003, ah.POINT_DTYPE, planted_from=train if n2 else None,
                                     nplanted=min(int(n1 * c["planted"]), n2), maxflip=c["maxflip"])
    query["_pad"] = 0xAB
    train["_pad"] = 0xCD
    return query, train


def run_case(ah, okz, torch, synth, det, c):
    query, train = make_sets(ah, synth, c)
    n1, n2 = len(query), len(train)
    want = okz.match(query.copy(), train)
    wk = query.copy()
    wl = okz.match_knn2(wk, train, c["knn2"][0], c["knn2"][1])
    fails = []
    d2 = torch.from_numpy(train.view(np.uint8).copy()).cuda() if n2 else torch.zeros(104, dtype=torch.uint8, device="cuda")
    ctx = det.ctx if c["ctx"] else None
    for kernel, env in (("mfma", {"HAK_MATCH_VALU": "0", "HAK_MATCH_QT": "1"}), ("mfma qt2", {"HAK_MATCH_VALU": "0", "HAK_MATCH_QT": "2"}), ("valu", {"HAK_MATCH_VALU": "1"})):
        os.environ.update(env)
        d1 = torch.from_numpy(query.view(np.uint8).copy()).cuda()
        got = query.copy()
        ah.check(ah.lib.hak_match(ctx, d1.data_ptr(), n1, d2.data_ptr(), n2, got.ctypes.data))
        dev = np.frombuffer(d1.cpu().numpy().tobytes(), ah.POINT_DTYPE)
        for f in MFIELDS:
            if not np.array_equal(got[f].view(np.uint32), want[f].view(np.uint32)):
                bad = np.nonzero(got[f].view(np.uint32) != want[f].view(np.uint32))[0]
                fails.append(f"{kernel} hak_match: field {f} differs at {bad.size} of {n1} queries (first {bad[:4].tolist()})")
            elif not np.array_equal(dev[f].view(np.uint32), want[f].view(np.uint32)):
                fails.append(f"{kernel} hak_match: device copy of field {f} differs from the host copy")
        # 2-NN + ratio + cross-check + compaction
        d1 = torch.from_numpy(query.view(np.uint8).copy()).cuda()
        hq = query.copy()
        d_out = torch.zeros(max(n1, 1) * 32, dtype=torch.uint8, device="cuda")
        h_out = np.zeros(max(n1, 1), ah.MATCH_PAIR_DTYPE)
        cnt = C.c_int(0)
        ah.check(ah.lib.hak_match_knn2(ctx, d1.data_ptr(), n1, d2.data_ptr(), n2, c["knn2"][0][0], c["knn2"][0][1], int(c["knn2"][1]), 0,
                                       hq.ctypes.data, d_out.data_ptr(), C.byref(cnt), h_out.ctypes.data))
        if cnt.value != len(wl):
            fails.append(f"{kernel} hak_match_knn2: {cnt.value} accepted, oracle {len(wl)}")
        else:
            for f in ah.MATCH_PAIR_DTYPE.names:
                if not np.array_equal(h_out[:cnt.value][f].view(np.uint32), wl[f].view(np.uint32)):
                    fails.append(f"{kernel} hak_match_knn2: list field {f} differs")
            for f in MFIELDS:
                if not np.array_equal(hq[f].view(np.uint32), wk[f].view(np.uint32)):
                    fails.append(f"{kernel} hak_match_knn2: point field {f} differs")
    for k in ("HAK_MATCH_VALU", "HAK_MATCH_QT"):
        os.environ.pop(k, None)
    return fails, int((want["match"] >= 0).sum()), len(wl)


def run(cases, seed, only=None, verbose=True, out=sys.stdout):
    import torch
    import akaze_hip as ah
    from akaze_hip import synth
    import okz
    okz.build()
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    okz.set_num_threads(max(1, min(16, ncpu)))
    det = ah.Akazer()
    det.init((320, 240, 384), max_pts=12000, batch=2)
    failed, ta, tk, tq, t0 = [], 0, 0, 0, time.time()
    idx = [only] if only is not None else range(cases)
    for i in idx:
        c = draw_case(seed, i)
        fails, na, nk = run_case(ah, okz, torch, synth, det, c)
        ta, tk, tq = ta + na, tk + nk, tq + c["n1"]
        if verbose or fails:
            print(f"{'FAIL' if fails else 'ok  '} #{i:<4d} {c['n1']:5d} x {c['n2']:<5d} planted={c['planted']} dup={c['dup']} maxflip={c['maxflip']} "
                  f"{'extremes ' if c['extremes'] else ''}{'ctx ' if c['ctx'] else 'pool '}knn2={c['knn2']}  [{na} accepted, {nk} 2-NN]", file=out, flush=True)
        for f in fails[:8]:
            print("       " + f, file=out, flush=True)
        if fails:
            failed.append(i)
    det.close()
    print(f"== seed {seed}: {len(list(idx))} cases x 3 kernels x 2 searches, {len(failed)} failed {failed}; {tq} queries, {ta} accepted 1-NN and {tk} accepted "
          f"2-NN matches per kernel, all compared with the oracle in {time.time() - t0:.0f} s", file=out, flush=True)
    return failed


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--only", type=int, default=None)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    sys.exit(1 if run(a.cases, a.seed, a.only, not a.quiet) else 0)
