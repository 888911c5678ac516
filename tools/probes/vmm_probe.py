"""What does hipMemSetAccess accept on this runtime?  (tools/vmm_lottery.py needs several physical handles in one range)"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from aircraftoptimalcontrol_amd import vmm
torch.cuda.init(); torch.zeros(1, device="cuda")
h = vmm._hip()
MB = 1 << 20
def acc():
    a = vmm._AccessDesc(); a.location.type, a.location.id, a.flags = 1, 0, 3
    return a
def attempt(name, sizes, how, align=0):
    base = C.c_void_p(0)
    tot = sum(sizes)
    rc = h.hipMemAddressReserve(C.byref(base), C.c_size_t(tot), C.c_size_t(align), None, C.c_ulonglong(0))
    res = ["reserve=%d base=%x" % (rc, base.value or 0)]
    off = 0
    prop = vmm._prop(0)
    hds = []
    for n in sizes:
        hd = C.c_void_p(0)
        r1 = h.hipMemCreate(C.byref(hd), C.c_size_t(n), C.byref(prop), C.c_ulonglong(0))
        r2 = h.hipMemMap(C.c_void_p(base.value + off), C.c_size_t(n), C.c_size_t(0), hd, C.c_ulonglong(0))
        a = acc()
        if how == "each":
            r3 = h.hipMemSetAccess(C.c_void_p(base.value + off), C.c_size_t(n), C.byref(a), C.c_size_t(1))
        elif how == "prefix":
            r3 = h.hipMemSetAccess(base, C.c_size_t(off + n), C.byref(a), C.c_size_t(1))
        else:
            r3 = -1
        res.append("create=%d map=%d access=%d" % (r1, r2, r3))
        hds.append((hd, off, n)); off += n
    if how == "end":
        a = acc()
        res.append("access_all=%d" % h.hipMemSetAccess(base, C.c_size_t(tot), C.byref(a), C.c_size_t(1)))
    # touch the memory
    try:
        t = torch.as_tensor(vmm._Raw(base.value, (tot // 8,), "<f8"), device="cuda:0")
        t.fill_(1.0); torch.cuda.synchronize()
        res.append("sum_ok=%s" % bool(t.sum().item() == tot // 8))
    except Exception as e:
        res.append("touch failed: %r" % (e,))
    torch.cuda.synchronize()
    for hd, o, n in hds:
        h.hipMemUnmap(C.c_void_p(base.value + o), C.c_size_t(n)); h.hipMemRelease(hd)
    h.hipMemAddressFree(base, C.c_size_t(tot))
    print(name, " | ".join(res), flush=True)

attempt("one handle 64M", [64 * MB], "each")
attempt("two handles, access each", [64 * MB, 64 * MB], "each")
attempt("two handles, access prefix", [64 * MB, 64 * MB], "prefix")
attempt("two handles, access once at the end", [64 * MB, 64 * MB], "end")
attempt("four 2M handles, end", [2 * MB] * 4, "end")
attempt("two handles 1G+1.5G, end", [1024 * MB, 1536 * MB], "end")
attempt("two handles, access each, align 2M", [64 * MB, 64 * MB], "each", 2 * MB)
