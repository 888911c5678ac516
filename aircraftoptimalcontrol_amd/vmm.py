"""Device memory with a placement the caller decides: HIP virtual-memory management behind a small arena.

Why (DESIGN.md section 4 "placement"): the duration of the write-heavy passes depends on which physical frames hold their
output streams, and torch's caching allocator hands a solver whatever `hipMalloc` returns.  An arena here is ONE reserved
virtual range (`hipMemAddressReserve`) backed by physical handles the caller sizes (`hipMemCreate` at the granularity
`hipMemGetAllocationGranularity` reports, mapped with `hipMemMap`): one handle for everything, one per buffer, or fixed-size
chunks, so that the layout of a solver's streams over physical memory is a property of the program and not of the
allocator's history.  The tensors it hands out are ordinary torch tensors over that memory (`__cuda_array_interface__`),
so nothing else in the host layer changes; the C-ABI takes plain pointers and never sees the difference.

Plumbing only: no numerical operation happens here, and the library itself still allocates nothing.
"""
import ctypes as C

import numpy as np

_HIP = None


def _hip():
    """The HIP runtime that is ALREADY mapped in this process (torch's): two runtimes in one process do not share
    devices or allocations (see _lib.lib()), so the library is opened by the path /proc/self/maps shows."""
    global _HIP
    if _HIP is None:
        import torch  # noqa: F401  (maps its libamdhip64)
        path = None
        try:
            for line in open("/proc/self/maps"):
                if "libamdhip64" in line:
                    path = line.split()[-1]
                    break
        except OSError:
            pass
        _HIP = C.CDLL(path or "libamdhip64.so")
    return _HIP


class _Location(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class _AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class _Prop(C.Structure):      # hipMemAllocationProp (hip_runtime_api.h)
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", _Location),
                ("win32HandleMetaData", C.c_void_p), ("allocFlags", _AllocFlags)]


class _AccessDesc(C.Structure):
    _fields_ = [("location", _Location), ("flags", C.c_int)]


def _check(rc, what):
    if rc != 0:
        h = _hip()
        h.hipGetErrorString.restype = C.c_char_p
        raise RuntimeError("%s: HIP error %d (%s)" % (what, rc, h.hipGetErrorString(rc).decode()))


def _prop(device_index):
    p = _Prop()
    p.type = 1                      # hipMemAllocationTypePinned
    p.requestedHandleType = 0       # hipMemHandleTypeNone
    p.location.type = 1             # hipMemLocationTypeDevice
    p.location.id = int(device_index)
    return p


def granularity(device_index=0, recommended=True):
    g = C.c_size_t(0)
    p = _prop(device_index)
    _check(_hip().hipMemGetAllocationGranularity(C.byref(g), C.byref(p), 1 if recommended else 0),
           "hipMemGetAllocationGranularity")
    return int(g.value)


class _Raw:
    """what torch.as_tensor() needs to see a device range as an array"""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


_TYPESTR = {"float64": "<f8", "float32": "<f4", "int32": "<i4", "uint8": "|u1"}


class Arena:
    """A reserved virtual range of `nbytes` on `device`, backed by physical handles as it is used: handles of `chunk_bytes`
    each (rounded up to the granularity), or — chunk_bytes None — one handle per take(new_handle=True), sized for that
    buffer (buffers taken without new_handle share the handle that is open, which grows by further handles of the
    buffer's size when it is full).  take(shape, dtype) carves tensors off the front.  The mapping lives as long as the
    arena object; tensors taken from it must not outlive it."""

    def __init__(self, device, nbytes, chunk_bytes=None, recommended=True, align=None, min_granule=2 << 20):
        import torch
        self.device = torch.device(device)
        self.index = self.device.index or 0
        torch.cuda.set_device(self.device)
        h = _hip()
        # the runtime reports 4 KiB on MI355X (ROCm 7.2), but a mapping that does not start on a 2 MiB boundary is refused
        # by hipMemSetAccess (invalid argument, measured): handles and offsets are kept to multiples of `min_granule`
        self.gran = max(granularity(self.index, recommended), int(min_granule))
        self._up = lambda v, a: (int(v) + a - 1) // a * a
        self.chunk = self._up(chunk_bytes, self.gran) if chunk_bytes else None
        self.size = self._up(nbytes, self.chunk or self.gran)
        self.base = C.c_void_p(0)
        _check(h.hipMemAddressReserve(C.byref(self.base), C.c_size_t(self.size), C.c_size_t(int(align or 0)), None, C.c_ulonglong(0)),
               "hipMemAddressReserve")
        self.handles = []      # (handle, offset, bytes)
        self.mapped = 0        # [0, mapped) is backed
        self.used = 0

    def _map(self, nbytes):
        """one more physical handle of nbytes (a multiple of the granularity) at the end of what is mapped"""
        h = _hip()
        if self.mapped + nbytes > self.size:
            raise MemoryError("arena of %d bytes exhausted (%d mapped, %d more asked)" % (self.size, self.mapped, nbytes))
        prop = _prop(self.index)
        hd = C.c_void_p(0)
        at = C.c_void_p(self.base.value + self.mapped)
        _check(h.hipMemCreate(C.byref(hd), C.c_size_t(nbytes), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
        _check(h.hipMemMap(at, C.c_size_t(nbytes), C.c_size_t(0), hd, C.c_ulonglong(0)), "hipMemMap")
        acc = _AccessDesc()
        acc.location.type, acc.location.id, acc.flags = 1, self.index, 3   # device, read-write
        _check(h.hipMemSetAccess(at, C.c_size_t(nbytes), C.byref(acc), C.c_size_t(1)), "hipMemSetAccess")
        self.handles.append((hd, self.mapped, nbytes))
        self.mapped += nbytes

    def take(self, shape, dtype="float64", zero=False, new_handle=False):
        import torch
        name = str(dtype).replace("torch.", "")
        item = np.dtype(_TYPESTR[name]).itemsize
        n = int(np.prod(shape)) * item
        start = self.mapped if new_handle else self._up(self.used, 256)
        if start < self.used:
            start = self._up(self.used, 256)
        end = start + n
        while self.mapped < end:
            self._map(self.chunk if self.chunk else self._up(end - self.mapped, self.gran))
        self.used = end
        t = torch.as_tensor(_Raw(self.base.value + start, shape, _TYPESTR[name]), device=self.device)
        if zero:
            t.zero_()
        return t

    def close(self):
        h = _hip()
        if self.base and self.base.value:
            import torch
            torch.cuda.synchronize(self.device)
            for hd, off, n in self.handles:
                h.hipMemUnmap(C.c_void_p(self.base.value + off), C.c_size_t(n))
                h.hipMemRelease(hd)
            h.hipMemAddressFree(self.base, C.c_size_t(self.size))
            self.base = C.c_void_p(0)
            self.handles = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
