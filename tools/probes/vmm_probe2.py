import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from aircraftoptimalcontrol_amd import vmm
torch.zeros(1, device="cuda")
h = vmm._hip()
MB = 1 << 20
def acc():
    a = vmm._AccessDesc(); a.location.type, a.location.id, a.flags = 1, 0, 3
    return a
def attempt(name, sizes):
    base = C.c_void_p(0)
    tot = sum(sizes)
    h.hipMemAddressReserve(C.byref(base), C.c_size_t(tot), C.c_size_t(0), None, C.c_ulonglong(0))
    off = 0
    prop = vmm._prop(0)
    hds = []
    for n in sizes:
        hd = C.c_void_p(0)
        h.hipMemCreate(C.byref(hd), C.c_size_t(n), C.byref(prop), C.c_ulonglong(0))
        h.hipMemMap(C.c_void_p(base.value + off), C.c_size_t(n), C.c_size_t(0), hd, C.c_ulonglong(0))
        a = acc()
        r = h.hipMemSetAccess(C.c_void_p(base.value + off), C.c_size_t(n), C.byref(a), C.c_size_t(1))
        hds.append((hd, off, n)); off += n
    res = []
    off = 0
    ts = []
    for i, n in enumerate(sizes):
        t = torch.as_tensor(vmm._Raw(base.value + off, (n // 8,), "<f8"), device="cuda:0")
        t.fill_(float(i + 1)); ts.append(t); off += n
    torch.cuda.synchronize()
    for i, t in enumerate(ts):
        res.append("part%d: min %.1f max %.1f" % (i, t.min().item(), t.max().item()))
    whole = torch.as_tensor(vmm._Raw(base.value, (tot // 8,), "<f8"), device="cuda:0")
    res.append("whole sum %.1f expected %.1f" % (whole.sum().item(), sum((i + 1) * (n // 8) for i, n in enumerate(sizes))))
    torch.cuda.synchronize()
    for hd, o, n in hds:
        h.hipMemUnmap(C.c_void_p(base.value + o), C.c_size_t(n)); h.hipMemRelease(hd)
    h.hipMemAddressFree(base, C.c_size_t(tot))
    print(name, " | ".join(res), flush=True)
attempt("2 x 64M", [64 * MB] * 2)
attempt("4 x 2M", [2 * MB] * 4)
attempt("3 x 1G", [1024 * MB] * 3)
attempt("1.5G + 1G + 1G", [1536 * MB, 1024 * MB, 1024 * MB])
