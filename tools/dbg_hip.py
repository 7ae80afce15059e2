import ctypes as C, os, sys
sys.path.insert(0, '.')
import torch
print("torch avail", torch.cuda.is_available(), torch.cuda.device_count(), torch.version.hip)
x = torch.zeros(4, device='cuda'); print(x.sum().item())
def maps():
    return sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'hip64' in l or 'hsa-runtime' in l))
print(maps())
L = C.CDLL(os.path.abspath('sigmod-2018_amd/librhj.so'))
print(maps())
for name in ('libamdhip64.so.7',):
    h = C.CDLL(name); n = C.c_int(-1); rc = h.hipGetDeviceCount(C.byref(n)); print(name, 'rc', rc, 'n', n.value)
print({k:v for k,v in os.environ.items() if 'VISIBLE' in k or 'ROC' in k or 'HIP' in k or 'HSA' in k})
