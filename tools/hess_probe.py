"""data-movement floor of the Hessian class's launch shapes (hak_op_hess_probe): GB/s of the 12 B/px compulsory bytes, per level of the demo
schedule (dilations 2, 3, 3, 4 in every octave), and the class total per launch sequence next to it"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import akaze_hip as ah
nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tot_ms = tot_b = 0.0
for (w, h, name) in ((1920, 1080, "octave 0"), (960, 540, "octave 1"), (480, 270, "octave 2"), (240, 135, "octave 3")):
    for step in (2, 3, 3, 4):
        ms, gbs = C.c_double(), C.c_double()
        ah.check(ah.lib.hak_op_hess_probe(w, h, nimg, step, 10, C.byref(ms), C.byref(gbs)))
        tot_ms += ms.value
        tot_b += 12.0 * w * h * nimg
        print(f"{name} {w}x{h} x {nimg}, dilation {step}: {ms.value * 1e3:8.1f} us  {gbs.value:7.1f} GB/s")
print(f"class total (16 launches): {tot_ms:.3f} ms for {tot_b / 1e9:.2f} GB compulsory = {tot_b / tot_ms / 1e6:.1f} GB/s")
