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
    """Device memory for the big streams of a solver, backed by physical handles the caller sizes.
      chunk_bytes = None: every take(new_handle=True) gets a reservation and ONE physical handle of its own, sized for that
                    buffer (takes without new_handle share the last one while it has room);
      chunk_bytes = n:    one reservation of `nbytes`, aligned to n, backed by uniform handles of n bytes mapped as the
                    arena is used (on this runtime a handle must sit at an offset that is a multiple of its size: 1.5 GiB +
                    1 GiB in one range is refused, equal chunks work — tools/probes/vmm_probe2.py).
    The mappings live as long as the arena object; tensors taken from it must not outlive it."""

    def __init__(self, device, nbytes, chunk_bytes=None, recommended=True, min_granule=2 << 20):
        import torch
        self.device = torch.device(device)
        self.index = self.device.index or 0
        torch.cuda.set_device(self.device)
        # the runtime reports 4 KiB on MI355X (ROCm 7.2); handles and offsets are kept to multiples of `min_granule`
        self.gran = max(granularity(self.index, recommended), int(min_granule))
        self._up = lambda v, a: (int(v) + a - 1) // a * a
        self.chunk = self._up(chunk_bytes, self.gran) if chunk_bytes else None
        self.size = self._up(nbytes, self.chunk or self.gran)
        self.segments = []     # [base, reserved bytes, mapped bytes, used bytes, [(handle, offset, bytes)]]
        if self.chunk:
            self._reserve(self.size, self.chunk)

    def _reserve(self, nbytes, align):
        base = C.c_void_p(0)
        _check(_hip().hipMemAddressReserve(C.byref(base), C.c_size_t(nbytes), C.c_size_t(int(align)), None, C.c_ulonglong(0)),
               "hipMemAddressReserve")
        self.segments.append([base.value, nbytes, 0, 0, []])
        return self.segments[-1]

    def _map(self, seg, nbytes):
        """one more physical handle of nbytes at the end of what is mapped in `seg`"""
        h = _hip()
        if seg[2] + nbytes > seg[1]:
            raise MemoryError("arena segment of %d bytes exhausted (%d mapped, %d more asked)" % (seg[1], seg[2], nbytes))
        prop = _prop(self.index)
        hd = C.c_void_p(0)
        at = C.c_void_p(seg[0] + seg[2])
        _check(h.hipMemCreate(C.byref(hd), C.c_size_t(nbytes), C.byref(prop), C.c_ulonglong(0)), "hipMemCreate")
        _check(h.hipMemMap(at, C.c_size_t(nbytes), C.c_size_t(0), hd, C.c_ulonglong(0)), "hipMemMap")
        acc = _AccessDesc()
        acc.location.type, acc.location.id, acc.flags = 1, self.index, 3   # device, read-write
        _check(h.hipMemSetAccess(at, C.c_size_t(nbytes), C.byref(acc), C.c_size_t(1)), "hipMemSetAccess")
        seg[4].append((hd, seg[2], nbytes))
        seg[2] += nbytes

    @property
    def handles(self):
        return [hd for seg in self.segments for hd in seg[4]]

    def take(self, shape, dtype="float64", zero=False, new_handle=False):
        import torch
        name = str(dtype).replace("torch.", "")
        item = np.dtype(_TYPESTR[name]).itemsize
        n = int(np.prod(shape)) * item
        if self.chunk:
            seg = self.segments[0]
            start = max(seg[2] if new_handle else 0, self._up(seg[3], 256))
            while seg[2] < start + n:
                self._map(seg, self.chunk)
        else:
            seg = self.segments[-1] if self.segments else None
            start = self._up(seg[3], 256) if seg else 0
            if new_handle or seg is None or start + n > seg[2]:
                seg = self._reserve(self._up(n, self.gran), self.gran)
                self._map(seg, seg[1])
                start = 0
        seg[3] = start + n
        t = torch.as_tensor(_Raw(seg[0] + start, shape, _TYPESTR[name]), device=self.device)
        if zero:
            t.zero_()
        return t

    def close(self):
        h = _hip()
        if self.segments:
            import torch
            torch.cuda.synchronize(self.device)
            for base, size, _, _, hds in self.segments:
                for hd, off, n in hds:
                    h.hipMemUnmap(C.c_void_p(base + off), C.c_size_t(n))
                    h.hipMemRelease(hd)
                h.hipMemAddressFree(C.c_void_p(base), C.c_size_t(size))
            self.segments = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
