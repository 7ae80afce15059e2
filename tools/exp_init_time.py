"""What the first device call costs: the HIP runtime's own initialisation (hipFree(0) through libamdhip64) and, after it,
librhj.so's context (stream, events, pinned block, kernel attributes = loading the code object)."""
import time, ctypes as C, os
t0 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
L = C.CDLL(os.path.abspath("sigmod-2018_amd/librhj.so"))
t1 = time.perf_counter()
hip.hipFree(None)
t2 = time.perf_counter()
L.rhj_dev_alloc.restype = C.c_void_p; L.rhj_dev_alloc.argtypes = [C.c_size_t]
L.rhj_dev_alloc(1024)
t3 = time.perf_counter()
print("dlopen %.1f ms, HIP runtime initialisation %.1f ms, librhj context %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
