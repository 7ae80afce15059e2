"""Partition alone (rhj_partition_device: one relation, 12-byte intermediates, 16-byte finals), for timing builds whose
pass 2 is wrong on purpose (SR_ABL): python tools/exp_part_abl.py <n> <bits> <lib.so> ...   (each build in its own process)"""
import json, subprocess, sys
CHILD = r'''
import importlib, ctypes as C, torch, sys, json
import numpy as np
sys.path.insert(0, ".")
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0, lib_path=sys.argv[3])
n, bits = int(sys.argv[1]), int(sys.argv[2])
x = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev)
x[:, 0] = torch.randint(-(1 << 62), 1 << 62, (n,), device=rhj.dev)
x[:, 1] = torch.arange(n, device=rhj.dev)
y = torch.empty_like(x)
rhj.set_bits(bits)
hist = np.zeros(1 << bits, dtype=np.uint64); psum = np.zeros(1 << bits, dtype=np.int64)
acc = {"ms_hist": [], "ms_scan": [], "ms_scatter": [], "ms_total": []}
for i in range(8):
    rhj.lib.rhj_partition_device(x.data_ptr(), n, y.data_ptr(), hist.ctypes.data_as(C.c_void_p), psum.ctypes.data_as(C.c_void_p))
    if i >= 2:
        st = rhj.stats()
        for k in acc: acc[k].append(st[k])
print("PJSON " + json.dumps({k: round(sorted(v)[len(v) // 2], 4) for k, v in acc.items()}))
'''
n, bits, libs = sys.argv[1], sys.argv[2], sys.argv[3:]
for rnd in range(2):
    for lib in libs:
        res = subprocess.run([sys.executable, "-c", CHILD, n, bits, lib], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=200)
        line = [l for l in res.stdout.decode().splitlines() if l.startswith("PJSON ")]
        print(lib.split("/")[-1], line[0][6:] if line else "FAILED " + res.stderr.decode()[-300:], flush=True)
