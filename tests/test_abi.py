"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/aoc.h declares, struct sizes agree, and argument errors are reported (no compute)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT
from aircraftoptimalcontrol_amd import _lib


def _declared():
    txt = open(os.path.join(ROOT, "include", "aoc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aoc_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    _lib.build_library()
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libaoc_hip.so does not export %s" % n
        assert n in _lib.SYMBOLS, "python binding misses %s" % n
    assert sorted(_lib.SYMBOLS) == names


def test_geometry_helpers_and_errors():
    lib = _lib.lib()
    assert lib.aoc_ntiles(1) == 1 and lib.aoc_ntiles(64) == 1 and lib.aoc_ntiles(65) == 2
    assert lib.aoc_tiled_elems(100, 500, 6) == 2 * 500 * 6 * 64
    up = lambda v: (v + 255) // 256 * 256
    base = lambda B: B * 500 * 16 * 8 + lib.aoc_spec_max() * B * 8 + up(lib.aoc_linesearch_scratch_bytes(B, 500))
    # a batch small enough to give every Armijo candidate a wavefront of its own (tiles x groups of three <= 256) keeps the
    # candidate trajectories: aoc_workspace_bytes(B) holds the largest store a batch of exactly B can ask for ...
    assert lib.aoc_workspace_bytes(64, 500) == base(64) + lib.aoc_candidate_bytes(64, 500, lib.aoc_spec_max())
    assert lib.aoc_candidate_bytes(64, 500, 10) == 10 * (64 * 500 * 24 + 64 * 4) + 16   # records (float32 states: 24 B), flags, stored count per tile
    for B, m in ((4096, 12), (8192, 6), (5440, 9), (4096, 10), (3264, 15)):    # (tiles, candidates) at the edge of the rule
        assert lib.aoc_default_ncand(B, m, m) == m, (B, m)
        assert lib.aoc_workspace_bytes(B, 500) >= base(B) + lib.aoc_candidate_bytes(B, 500, m), (B, m)
    assert lib.aoc_default_ncand(8192 + 64, 6, 6) == 0 and lib.aoc_default_ncand(4096, 9, 10) == 0
    # ... a large batch none (three solvers of 131 072 trajectories used to carry 3.5 GB they never touched) ...
    assert lib.aoc_workspace_bytes(131072, 500) == base(131072)
    # ... and aoc_solve_workspace_bytes the largest store of ANY batch up to B: aoc_newton_solve runs its re-packed,
    # smaller generations in the workspace of the first (768 stored (tile, candidate) pairs at most)
    extra = max(lib.aoc_workspace_bytes(64 * t, 500) - base(64 * t) for t in range(1, 257))
    assert lib.aoc_solve_workspace_bytes(131072, 500) - lib.aoc_solve_workspace_bytes(131072, 500) % 256 >= base(131072) + extra - 256
    assert 768 * (500 * 64 * 24 + 256) <= extra < 768 * (500 * 64 * 24 + 256) + 4096
    assert b"gfx950" in lib.aoc_version()
    assert lib.aoc_strerror(-1) == b"invalid argument"
    # struct layout must match the header: 9 doubles + 76 doubles + 8 int32 + 2 pointers
    assert C.sizeof(_lib.Model) == 72
    assert C.sizeof(_lib.Problem) == 72 + 76 * 8 + 32 + 16
    assert C.sizeof(_lib.Params) == 48
    assert C.sizeof(_lib.Tuning) == 104   # 26 int32 (round 5: fw_wpe1, hcut_chain6, bw_hcut_full, fw_duo, hcut_waves, hcut_pairs)
    assert C.sizeof(_lib.MpcNoise) == 64  # uint64 seed, two uint32, six doubles (ABI revision 5)
    assert lib.aoc_abi_version() == _lib.AOC_ABI_VERSION == 5
    # argument errors are reported before anything touches a device
    p = _lib.Problem()
    assert lib.aoc_traj_cost(C.byref(p), None, None, None, None) == -1
    assert lib.aoc_pack(0, 10, 6, None, None, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "_SO", "/nonexistent/libaoc_hip.so")
    with pytest.raises(_lib.AocError):
        _lib.lib()


def test_default_speculation_depth_is_host_logic(tuned):
    """aoc_default_nspec: how many Armijo candidates ride along in the forward pass, by batch size (tiles x
    workgroups of three candidates <= 512), capped by armijo_maxiters and aoc_spec_max(); no device involved."""
    lib = _lib.lib()
    assert lib.aoc_spec_max() >= 10
    tuned(nspec=0)
    # where every candidate rides along, so does the step an exhausted search applies (index armijo_maxiters): m + 1
    for B, m, want in ((1, 10, 11), (4096, 10, 11), (8192, 10, 11), (8193, 10, 9), (10880, 10, 9), (10881, 10, 6),
                       (16384, 10, 6), (16385, 10, 2), (131072, 10, 2), (64, 20, 2), (64, 1, 1), (64, 2, 2),
                       (64, 4, 5), (64, 12, 13), (64, 15, 15), (64, 14, 15)):
        assert lib.aoc_default_nspec(B, m) == want, (B, m, lib.aoc_default_nspec(B, m), want)
    tuned(nspec=3)
    assert lib.aoc_default_nspec(131072, 10) == 3


def test_tuning_is_read_once_and_overridable():
    """The scheduling knobs come from the environment ONCE (first use) and afterwards only through
    aoc_set_tuning(); aoc_set_tuning(NULL) goes back to the defaults."""
    import os
    lib = _lib.lib()
    t0 = _lib.Tuning()
    lib.aoc_get_tuning(C.byref(t0))
    os.environ["AOC_SPLIT_TILES"] = str(t0.split_tiles + 7)       # too late: already initialised
    try:
        t1 = _lib.Tuning()
        lib.aoc_get_tuning(C.byref(t1))
        assert t1.split_tiles == t0.split_tiles
        with _lib.tuning(split_tiles=3, ls_cpl=2) as t:
            assert (t.split_tiles, t.ls_cpl) == (3, 2)
            lib.aoc_get_tuning(C.byref(t1))
            assert (t1.split_tiles, t1.ls_cpl, t1.ls_wcap) == (3, 2, t0.ls_wcap)
        lib.aoc_get_tuning(C.byref(t1))
        assert bytes(t1) == bytes(t0)
        lib.aoc_set_tuning(None)                                      # defaults = the environment as it is NOW
        lib.aoc_get_tuning(C.byref(t1))
        assert t1.split_tiles == t0.split_tiles + 7
    finally:
        os.environ.pop("AOC_SPLIT_TILES")
        lib.aoc_set_tuning(C.byref(t0))


def test_argument_errors_carry_a_reason():
    lib = _lib.lib()
    p = _lib.Problem()
    p.B, p.T, p.ref = 4, 10, 1
    p.RRt[1], p.RRt[2] = 1.0, 2.0
    assert lib.aoc_traj_cost(C.byref(p), 1, 1, 1, 1) == -1
    assert b"not symmetric" in lib.aoc_last_hip_error()
    p.RRt[2] = 1.0
    p.T = 2
    assert lib.aoc_traj_cost(C.byref(p), 1, 1, 1, 1) == -1
    assert b"T = 2" in lib.aoc_last_hip_error()


def test_too_small_a_workspace_is_an_error_not_a_fault():
    """aoc_newton_iterate / aoc_newton_solve / aoc_mpc_step are told the size of their workspace, aoc_forward /
    aoc_linesearch* the sizes of their scratch and candidate regions, and refuse one that is too small before anything
    is launched."""
    lib = _lib.lib()
    p = _lib.Problem()
    p.B, p.T, p.ref = 4096, 500, 1
    prm = _lib.Params(200, 10, 1.0, 0.5, 0.7, -1e-6, 8, 0)
    need = lib.aoc_workspace_bytes(4096, 500)
    rc = lib.aoc_newton_iterate(C.byref(p), C.byref(prm), 0, 1, 1, 1, 1, 1, need // 100, 1, 1, 1, 1, 1, 1, 1)
    assert rc == -1 and b"workspace" in lib.aoc_last_hip_error()
    rc = lib.aoc_newton_solve(C.byref(p), C.byref(prm), 1, 1, 1, 1, 1000, 4, 1, 1, 1, 1, 1, None, None, None, None, None)
    assert rc == -1 and b"workspace" in lib.aoc_last_hip_error()
    # pass level: line-search scratch and candidate store
    sb, cb = lib.aoc_linesearch_scratch_bytes(4096, 500), lib.aoc_candidate_bytes(4096, 500, 10)
    assert lib.aoc_linesearch_search(C.byref(p), C.byref(prm), 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, sb - 1) == -1
    assert b"scratch" in lib.aoc_last_hip_error()
    assert lib.aoc_linesearch_update(C.byref(p), C.byref(prm), 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, sb - 1, 0, None, None, 0) == -1
    assert b"scratch" in lib.aoc_last_hip_error()
    assert lib.aoc_linesearch_update(C.byref(p), C.byref(prm), 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, sb, 10, 1, 1, cb - 1) == -1
    assert b"candidate" in lib.aoc_last_hip_error()
    assert lib.aoc_linesearch(C.byref(p), C.byref(prm), 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 16, None, 0) == -1
    assert b"scratch" in lib.aoc_last_hip_error()
    assert lib.aoc_forward(C.byref(p), C.byref(prm), 10, 1, 1, 1, 1, 1, 1, 1, 1, 1, cb - 1, None) == -1
    assert b"candidate" in lib.aoc_last_hip_error()
    # a candidate store sized for another n_spec (the overrun of round 2) is refused too
    assert lib.aoc_forward(C.byref(p), C.byref(prm), 10, 1, 1, 1, 1, 1, 1, 1, 1, 1, lib.aoc_candidate_bytes(4096, 500, 7), None) == -1
