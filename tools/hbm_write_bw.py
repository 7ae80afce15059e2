import torch, time
x = torch.empty((100_000_000, 2), dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
def t(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n
print("fill 1.6 GB: %.3f ms" % t(lambda: x.fill_(7)))
print("copy 1.6 GB -> 1.6 GB: %.3f ms" % t(lambda: y.copy_(x)))
print("read-sum 1.6 GB: %.3f ms" % t(lambda: x.sum()))
