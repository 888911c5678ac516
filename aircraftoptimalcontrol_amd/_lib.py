"""ctypes binding of libaoc_hip.so (C-ABI declared in include/aoc.h).

There is no CPU fallback: if the shared library has not been built, or no GPU is visible when a
compute entry point is called, an AocError is raised.
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("AOC_LIB") or os.path.join(_PKG, "lib", "libaoc_hip.so")  # AOC_LIB: experiment builds
_SRC = [os.path.join(_PKG, "csrc", f) for f in ("aoc_kernels.hip", "aoc_device.h", "aoc_passes.inc")]   # [0]: the translation unit
_SRC += sorted(os.path.join(_PKG, "csrc", "passes", f) for f in os.listdir(os.path.join(_PKG, "csrc", "passes")) if f.endswith(".inc"))
_HDR = os.path.join(os.path.dirname(_PKG), "include", "aoc.h")

AOC_TILE = 64
AOC_ABI_VERSION = 5   # include/aoc.h: the revision this binding (struct layouts, argument lists) is written against

# status flags (include/aoc.h)
ST_NAN, ST_VNONPOS, ST_SINGULAR, ST_REGULARISED, ST_ARMIJO_EXH, ST_CONVERGED = 1, 2, 4, 8, 16, 32


class AocError(RuntimeError):
    pass


def library_path():
    return _SO


def build_library(force=False, verbose=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    deps = _SRC + [_HDR]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return _SO
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", _SRC[0], "-o", _SO]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return _SO


class Model(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("cd0", "cda", "cla", "m", "g", "S", "rho", "J", "dt")]


class Problem(C.Structure):
    _fields_ = [("model", Model), ("QQt", C.c_double * 36), ("RRt", C.c_double * 4), ("QQT", C.c_double * 36),
                ("B", C.c_int32), ("T", C.c_int32), ("x_in_f32", C.c_int32), ("x_out_f32", C.c_int32),
                ("ref_per_traj", C.c_int32), ("ref_T", C.c_int32), ("x_is_rollout", C.c_int32), ("reserved", C.c_int32),
                ("ref", C.c_void_p), ("stream", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("armijo_maxiters", C.c_int32), ("stepsize_0", C.c_double),
                ("cc", C.c_double), ("beta", C.c_double), ("term_cond", C.c_double),
                ("hessian_switch", C.c_int32), ("reserved", C.c_int32)]


class MpcNoise(C.Structure):
    """aoc_mpc_noise (include/aoc.h): the disturbance model aoc_mpc_step draws on the device."""
    _fields_ = [("seed", C.c_uint64), ("step", C.c_uint32), ("first", C.c_uint32), ("sigma", C.c_double * 6)]


class Tuning(C.Structure):
    """aoc_tuning (include/aoc.h): scheduling knobs; results never depend on them."""
    _fields_ = [(n, C.c_int32) for n in ("nspec", "split_tiles", "split_bw_tiles", "fw_lin", "ls_wcap", "ls_kgrow",
                                         "trial_split", "solve_norepack", "ls_worklist", "ls_cpl", "ls_depth_min",
                                         "fw_recompute", "store_candidates", "bw4_tiles", "bw5", "solve_repack_pct",
                                         "solve_sync_fast", "solve_split_tiles", "track_hcut", "bw_hcut", "fw_wpe1",
                                         "hcut_chain6", "bw_hcut_full", "fw_duo", "hcut_waves", "hcut_pairs")]


# every symbol include/aoc.h declares: (name, restype, argtypes)
_P, _I, _D, _Z = C.c_void_p, C.c_int32, C.c_double, C.c_size_t
SYMBOLS = {
    "aoc_get_tuning": (None, [_P]),
    "aoc_set_tuning": (None, [_P]),
    "aoc_version": (C.c_char_p, []),
    "aoc_strerror": (C.c_char_p, [C.c_int]),
    "aoc_last_hip_error": (C.c_char_p, []),
    "aoc_tiled_elems": (_Z, [_I, _I, _I]),
    "aoc_ntiles": (_I, [_I]),
    "aoc_pack": (C.c_int, [_I, _I, _I, _P, _P, _P]),
    "aoc_unpack": (C.c_int, [_I, _I, _I, _P, _P, _P]),
    "aoc_pack_f32": (C.c_int, [_I, _I, _I, _P, _P, _P]),
    "aoc_unpack_f32": (C.c_int, [_I, _I, _I, _P, _P, _P]),
    "aoc_step_batch": (C.c_int, [_P, _I] + [_P] * 10),
    "aoc_cost_batch": (C.c_int, [_P, _I] + [_P] * 10),
    "aoc_traj_cost": (C.c_int, [_P, _P, _P, _P, _P]),
    "aoc_initial_trajectory": (C.c_int, [_P, _D, _D, _P, _P, _P]),
    "aoc_rollout_cost": (C.c_int, [_P] * 9),
    "aoc_backward": (C.c_int, [_P, _I] + [_P] * 7 + [_Z]),
    "aoc_backward_scratch_bytes": (_Z, [_I, _I]),
    "aoc_gradient": (C.c_int, [_P] * 7),
    "aoc_forward": (C.c_int, [_P, _P, _I] + [_P] * 9 + [_Z, _P]),
    "aoc_candidate_bytes": (_Z, [_I, _I, _I]),
    "aoc_default_ncand": (_I, [_I, _I, _I]),
    "aoc_linesearch_scratch_bytes": (_Z, [_I, _I]),
    "aoc_spec_max": (_I, []),
    "aoc_default_nspec": (_I, [_I, _I]),
    "aoc_linesearch": (C.c_int, [_P, _P, _I] + [_P] * 13 + [_Z, _P, _Z]),
    "aoc_linesearch_search": (C.c_int, [_P, _P, _I] + [_P] * 9 + [_Z]),
    "aoc_linesearch_update": (C.c_int, [_P] * 12 + [_Z, _I, _P, _P, _Z]),
    "aoc_lqr_tracking": (C.c_int, [_P] * 9),
    "aoc_ltv_lqr": (C.c_int, [_I, _I, _I] + [_P] * 17),
    "aoc_workspace_bytes": (_Z, [_I, _I]),
    "aoc_newton_iterate": (C.c_int, [_P, _P, _I] + [_P] * 5 + [_Z] + [_P] * 7),
    "aoc_solve_workspace_bytes": (_Z, [_I, _I]),
    "aoc_newton_solve": (C.c_int, [_P] * 6 + [_Z, _I] + [_P] * 10),
    "aoc_newton_solve2": (C.c_int, [_P] * 6 + [_Z, _I] + [_P] * 11),
    "aoc_abi_version": (_I, []),
    "aoc_summary": (C.c_int, [_I, _P, _P, _P, _P, _I, _P]),
    "aoc_streams_concurrent": (C.c_int, [_P, _P]),
    "aoc_solve_trace": (C.c_int, [_P, _I]),
    "aoc_solve_trace_rows": (_I, []),
    "aoc_mpc_step": (C.c_int, [_P] * 3 + [_I] + [_P] * 6 + [_Z] + [_P] * 16),
    "aoc_traj_cost_f32": (C.c_int, [_P] * 5),
    "aoc_initial_trajectory_f32": (C.c_int, [_P, _D, _D, _P, _P, _P]),
    "aoc_rollout_cost_f32": (C.c_int, [_P] * 9),
    "aoc_workspace_bytes_f32": (_Z, [_I, _I]),
    "aoc_newton_iterate_f32": (C.c_int, [_P, _P, _I] + [_P] * 5 + [_Z] + [_P] * 7),
}

_lib = None


def lib():
    """The loaded C library.  Raises AocError (never falls back) if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise AocError("libaoc_hip.so is not built (%s). Run __graft_entry__.build() or "
                           "aircraftoptimalcontrol_amd.build_library(); there is no CPU fallback." % _SO)
        # torch-ROCm wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
        # process do not share devices or allocations, so torch (whose allocator owns the buffers
        # we are handed) must be loaded first: the dynamic linker then resolves our NEEDED
        # libamdhip64.so.7 to the copy that is already mapped.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        try:
            l = C.CDLL(_SO)
        except OSError as e:  # e.g. ROCm runtime missing
            raise AocError("cannot load %s: %s" % (_SO, e))
        try:
            l.aoc_abi_version.restype = C.c_int32
            have = int(l.aoc_abi_version())
        except AttributeError:
            have = None
        if have != AOC_ABI_VERSION:   # an older or newer library would be handed shifted arguments
            raise AocError("%s has ABI revision %s, this binding is written against %d (include/aoc.h AOC_ABI_VERSION): "
                           "rebuild the library (__graft_entry__.build())" % (_SO, have, AOC_ABI_VERSION))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class tuning:
    """Context manager over aoc_set_tuning(): `with tuning(nspec=2, ls_worklist=1): ...` overrides the named
    knobs for the block and restores the previous settings afterwards (tests, tuning tools)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        l = lib()
        self.old = Tuning()
        l.aoc_get_tuning(C.byref(self.old))
        new = Tuning.from_buffer_copy(self.old)
        for k, v in self.kw.items():
            if k not in dict(Tuning._fields_) or k.startswith("reserved"):
                raise AttributeError("aoc_tuning has no field %r" % k)
            setattr(new, k, int(v))
        l.aoc_set_tuning(C.byref(new))
        return new

    def __exit__(self, *exc):
        lib().aoc_set_tuning(C.byref(self.old))
        return False


def check(rc, what=""):
    if rc != 0:
        l = lib()
        raise AocError("%s failed: %s (%s)" % (what or "aoc call", l.aoc_strerror(rc).decode(),
                                                l.aoc_last_hip_error().decode()))


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise AocError("no GPU visible: the aoc HIP path has no CPU fallback")
    return torch
