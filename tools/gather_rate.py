"""random-sector ceiling of the box: k_gather_probe (one dword per lane from pseudo-random 128-byte lines of a 4 GiB buffer, every
load of a lane independent) at several grid sizes and loads in flight; GB/s if every gather moves one 64-byte sector
(profiles/r02_gather_calib.txt: FETCH_SIZE charges 63.9 B per gather).  What the descriptor stage's gathers could reach if nothing
but HBM's random-access rate held them back."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import akaze_hip as ah
for blocks in (1024, 2048, 4096, 8192, 16384):
    for per_lane in (16, 64):
        ms = C.c_double()
        ah.check(ah.lib.hak_op_gather_probe(4 << 30, blocks, per_lane, 4, C.byref(ms)))
        n = blocks * 256 * per_lane
        print(f"{blocks:6d} blocks x 256 lanes x {per_lane:3d} gathers: {ms.value:8.3f} ms  {n / ms.value / 1e6:8.2f} G gathers/s  {n * 64 / ms.value / 1e6:8.1f} GB/s of 64-byte sectors")
