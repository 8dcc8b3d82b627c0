"""prints every launch shape of the copy-ceiling probe (hak_op_copy_probe_shapes) on this box: GB/s of read + write"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-akaze_amd"))
import akaze_hip as ah
out = (C.c_double * 26)()
ah.check(0 if ah.lib.hak_op_copy_probe_shapes(int(sys.argv[1]) if len(sys.argv) > 1 else 2 << 30, 10, out, 26) == 26 else 1)
grids = ["8 blk/CU", "16 blk/CU", "32 blk/CU", "one pass"]
for g in range(4):
    for st in range(2):
        print(f"{grids[g]:10s} {'nt   ' if st == 0 else 'plain'} stores: " + "  ".join(f"NI={2 << l}: {out[g * 6 + st * 3 + l]:7.1f}" for l in range(3)))
print(f"read only {out[24]:7.1f} GB/s   write only {out[25]:7.1f} GB/s   best copy {max(out[:24]):7.1f} GB/s")
