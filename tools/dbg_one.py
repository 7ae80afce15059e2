import importlib, ctypes as C, torch, sys
sys.path.insert(0, ".")
import bench
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
n, bits = int(sys.argv[1]), int(sys.argv[2])
w = dict(nR=n, nS=n, bits=bits, dist="uniform")
rhj.set_bits(bits)
R, S = bench.make_relations(w, rhj.dev, 7)
out = torch.empty((n, 2), dtype=torch.int64, device=rhj.dev)
m = C.c_uint64(0)
torch.cuda.synchronize()
rc = rhj.lib.rhj_join_device(R.data_ptr(), n, S.data_ptr(), n, out.data_ptr(), n, C.byref(m))
print("rc", rc, "m", m.value, rhj.stats())
