// one_kernel.hip — compile ONE kernel of the library (fp64 build; -DARITH32: the float32 build) for register / ISA studies, in seconds instead of the
// two minutes of the whole library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 --offload-device-only -S -Rpass-analysis=kernel-resource-usage \
//         -DKERNEL='k_backward<true, false, true, true, float>' tools/probes/one_kernel.hip -o /tmp/k.s
// (tools/one_kernel.sh wraps this and prints the resource line.)
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/aoc.h"

namespace aoc_common {
constexpr int TILE = 64;
}
#define AOC_DEVICE_COMMON
using aoc_common::TILE;
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
extern "C" int32_t aoc_ntiles(int32_t B) { return (B + TILE - 1) / TILE; }
constexpr int AOC_SPEC_MAX = 15;

#define AOC_KERNELS_ONLY
#ifdef ARITH32   // the float32 build (aoc32): -DARITH32
#define AOC_ARITH_NS aoc32
#define AOC_REAL float
#define PROBE_NS aoc32
namespace aoc64 { struct LsState { int pad[8]; double alpha[64]; }; }   // only its size is used (ls_carve is host code)
#else
#define AOC_ARITH_NS aoc64
#define AOC_REAL double
#define PROBE_NS aoc64
#endif
#include "../../aircraftoptimalcontrol_amd/csrc/aoc_device.h"
#include "../../aircraftoptimalcontrol_amd/csrc/aoc_passes.inc"

#ifndef KERNEL
#define KERNEL k_backward<true, false, true, true, float>
#endif
// taking the address instantiates the kernel
extern "C" const void* aoc_probe_kernel() { return (const void*)&PROBE_NS::KERNEL; }
