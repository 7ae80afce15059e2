"""Phase stamps of pass 2 (k_scatter_runs, diagnostics build): python tools/exp_sr_stamps.py [n] [bits]   (RHJ_LIB=.../librhj_instr.so)
slots: 0 loop top | 1 after barrier (loads landed? no: counters zero) | 2 ranks done | 3 barrier | 4 digit prefixes + barrier | 5 staged |
6 next run table | 7 barrier | 8 next loads issued | 9 written out + counters zeroed"""
import importlib, ctypes as C, torch, sys, json
import numpy as np
sys.path.insert(0, ".")
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 12
x = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev)
x[:, 0] = torch.randint(-(1 << 62), 1 << 62, (n,), device=rhj.dev)
x[:, 1] = torch.arange(n, device=rhj.dev)
y = torch.empty_like(x)
rhj.set_bits(bits)
hist = np.zeros(1 << bits, dtype=np.uint64); psum = np.zeros(1 << bits, dtype=np.int64)
for i in range(3):
    rhj.lib.rhj_partition_device(x.data_ptr(), n, y.data_ptr(), hist.ctypes.data_as(C.c_void_p), psum.ctypes.data_as(C.c_void_p))
buf = np.zeros(256 * 16, dtype=np.uint64)
rhj.lib.rhj_debug_sr_stamps.argtypes = [C.c_void_p]
assert rhj.lib.rhj_debug_sr_stamps(buf.ctypes.data_as(C.c_void_p)) == 0
s = buf.reshape(256, 16).astype(np.int64)
d = (s[:, 1:10] - s[:, 0:9]) * 10      # ns per phase
wait = (s[:, 10] - s[:, 1]) * 10
names = ["0 barrier(top)", "1 rank", "2 barrier", "3 prefix+barrier", "4 stage", "5 next-run-table", "6 barrier", "7 issue loads", "8 write-out+zero"]
print(json.dumps({"load_wait_inside_rank_ns_median": int(np.median(wait)), "stage_ms": rhj.stats()["ms_scatter"], "batch_ns_median": int(np.median(s[:, 9] - s[:, 0]) * 10),
                  "phases_ns_median": {nm: int(np.median(d[:, i])) for i, nm in enumerate(names)},
                  "phases_ns_p90": {nm: int(np.percentile(d[:, i], 90)) for i, nm in enumerate(names)}}))
