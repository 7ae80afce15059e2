#!/bin/bash
# timing experiments: join-phase time with parts removed (results are wrong by design)
# needs the diagnostics build: make -C sigmod-2018_amd instr
export RHJ_LIB=${RHJ_LIB:-sigmod-2018_amd/librhj_instr.so}
for m in 0 1 2 3; do
  RHJ_ABLATE=$m python3 - <<PY
import importlib, ctypes as C, torch, sys
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
w = bench.WORKLOADS["c3"]; rhj.set_bits(w["bits"])
R, S = bench.make_relations(w, rhj.dev, 1234)
out = torch.empty((w["nS"], 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
for i in range(3):
    rhj.lib.rhj_join_device(R.data_ptr(), w["nR"], S.data_ptr(), w["nS"], out.data_ptr(), w["nS"], C.byref(m))
st = rhj.stats()
print("ablate $m: probe %.3f ms count %.3f build %.3f  matches %d" % (st["ms_probe"], st["ms_count"], st["ms_build"], m.value))
PY
done
