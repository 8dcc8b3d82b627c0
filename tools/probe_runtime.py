import sys, os, subprocess
sc = sys.argv[1] if len(sys.argv) > 1 else None
if sc is None:
    for s in ("lib_only", "torch_first", "lib_first"):
        r = subprocess.run([sys.executable, __file__, s], capture_output=True, text=True)
        print("=====", s, "rc", r.returncode); print(r.stdout[-3000:]); print(r.stderr[-1500:])
    sys.exit(0)
sys.path[:0] = ['cuda-akaze_amd', 'oracle']
import ctypes as C
def maps():
    for l in open('/proc/self/maps'):
        if 'libamdhip64' in l or 'libhsa-runtime' in l:
            print('   ', l.split()[-1])
            
if sc == "lib_only":
    import akaze_hip as ah
    print("count", ah.device_count()); 
    p = C.c_void_p(); pitch = C.c_int()
    print("alloc", ah.lib.hak_image_alloc(C.byref(p), 256, 128, C.byref(pitch)), ah.lib.hak_last_error())
    q = C.c_void_p(); ah.lib.hak_image_alloc(C.byref(q), 256, 128, C.byref(pitch))
    print("lowpass", ah.lib.hak_op_lowpass(p, q, 256, 128, pitch.value, 1.0, 2), ah.lib.hak_last_error())
    print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l)))
elif sc == "torch_first":
    import torch; print("torch cuda", torch.cuda.is_available()); x = torch.zeros(256*128, device='cuda'); y = torch.zeros_like(x)
    import akaze_hip as ah
    print("count", ah.device_count())
    print("lowpass", ah.lib.hak_op_lowpass(x.data_ptr(), y.data_ptr(), 256, 128, 256, 1.0, 2), ah.lib.hak_last_error())
    print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l)))
else:
    import akaze_hip as ah
    print("count before torch", ah.device_count())
    import torch; print("torch cuda", torch.cuda.is_available()); x = torch.ones(256*128, device='cuda'); y = torch.zeros_like(x)
    print("count", ah.device_count())
    print("lowpass", ah.lib.hak_op_lowpass(x.data_ptr(), y.data_ptr(), 256, 128, 256, 1.0, 2), ah.lib.hak_last_error())
    print(y[:4])
    print(sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l or 'libhsa-runtime' in l)))
