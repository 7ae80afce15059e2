"""What the chip delivers for the access pattern of the fused join's phase 1, without the join: every workgroup
(1024 threads, 4 loads in flight per lane, sc1 buffer loads) gathers random 16-byte tuples from its own contiguous
region.  Needs the diagnostics build (make -C sigmod-2018_amd instr).  Prints G gathers/s for region sizes from
L2-resident to C3's 24.4 K tuples per workgroup, with and without a probe-like stream next to the gathers."""
import ctypes as C, importlib, json, os, sys
sys.path.insert(0, ".")
os.environ.setdefault("RHJ_LIB", os.path.join("sigmod-2018_amd", "librhj_instr.so"))
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
f = rhj.lib.rhj_debug_gather_bench
f.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
out = []
for wgs in (256, 512):
    for region in (2048, 6104, 12207, 24414, 65536):
        for stream in (0, 1):
            ms = C.c_float(0)
            rounds = 96
            assert f(region, rounds, stream, wgs, C.byref(ms)) == 0
            gathers = wgs * 1024 * 4 * rounds
            out.append({"wgs": wgs, "region_tuples": region, "live_MB": round(wgs * region * 16 / 1e6, 1), "stream": stream,
                        "ms": round(ms.value, 4), "G_gathers_per_s": round(gathers / ms.value / 1e6, 1)})
            print(json.dumps(out[-1]))
