"""data-movement floor of the FED family's launch shapes (hak_op_stream_probe) next to the measured kernels: GB/s of compulsory bytes"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import akaze_hip as ah
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 384
for (w, h, name) in ((1920, 1080, "octave 0"), (960, 540, "octave 1"), (480, 270, "octave 2"), (240, 135, "octave 3")):
    for nw, warm, what in ((2, 7, "k_fed_sf<3>: read L, write smooth + L'"), (3, 8, "k_fed_sf<4> + g"), (1, 6, "k_fed_multi<3>: (read L + g ~) read 1, write 1")):
        ms, gbs = C.c_double(), C.c_double()
        ah.check(ah.lib.hak_op_stream_probe(w, h, nimg, nw, warm, 10, C.byref(ms), C.byref(gbs)))
        print(f"{name} {w}x{h} x {nimg}: 1 read + {nw} write planes, {warm} warm-up rows: {ms.value * 1e3:8.1f} us  {gbs.value:7.1f} GB/s   ({what})")
