"""Filter scan throughput on a device-resident column (algorithmic bytes 8n + 8 hits)."""
import importlib, sys, json, ctypes as C
import torch
sys.path.insert(0, ".")
mod = importlib.import_module("sigmod-2018_amd"); rhj = mod.RHJ(device=0)
out = {}
n = 400_000_000
col = torch.randint(0, 1 << 20, (n,), dtype=torch.int64, device=rhj.dev)
dst = torch.empty(n, dtype=torch.int64, device=rhj.dev)
hits = C.c_uint64(0)
for name, op, v in (("sel50", ">", 1 << 19), ("sel1", "<", 10486), ("eq", "=", 12345)):
    for i in range(3):
        rc = rhj.lib.rhj_filter_device(col.data_ptr(), None, n, op.encode(), v, dst.data_ptr(), C.byref(hits))
        assert rc == 0
    ms = rhj.stats()["ms_total"]
    want = int(((col > v) if op == ">" else (col < v) if op == "<" else (col == v)).sum())
    assert hits.value == want
    idx = dst[:hits.value]
    assert bool((idx[1:] > idx[:-1]).all())
    b = 8 * n + 8 * hits.value
    out[name] = {"n": n, "hits": hits.value, "ms": ms, "GBps": b / ms / 1e6, "frac_of_8TBps": b / ms / 1e6 / 8000}
sel = torch.randint(0, n, (100_000_000,), dtype=torch.int64, device=rhj.dev)
for i in range(3):
    rhj.lib.rhj_filter_device(col.data_ptr(), sel.data_ptr(), sel.shape[0], b">", 1 << 19, dst.data_ptr(), C.byref(hits))
ms = rhj.stats()["ms_total"]
out["indirect_sel50"] = {"n": sel.shape[0], "hits": hits.value, "ms": ms, "GBps": (16 * sel.shape[0] + 8 * hits.value) / ms / 1e6}
print(json.dumps(out))
