import importlib, ctypes as C, torch, sys, os
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
nR, nS = [int(x) for x in sys.argv[1:3]]
w = dict(nR=nR, nS=nS, bits=12, dist="uniform")
rhj.set_bits(12)
R, S = bench.make_relations(w, rhj.dev, 1234)
cap = max(nR, nS)
out = torch.empty((cap, 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
for i in range(3):
    rhj.lib.rhj_join_device(R.data_ptr(), nR, S.data_ptr(), nS, out.data_ptr(), cap, C.byref(m))
st = rhj.stats()
print("ABL=%s nR=%d nS=%d: fused %.3f ms matches %d" % (os.environ.get("RHJ_ABLATE", "0"), nR, nS, st["ms_probe"], m.value))
