// Probe: what a copy (read + write) can reach on this box, by access policy and bytes in flight per thread — the
// ceiling the passes (which all read AND write) are compared with.  4 GiB source, 4 GiB destination.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/copy_probe tools/probes/hbm_copy_probe.hip && /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U, bool NT_LD, bool NT_ST>
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ p, double2* __restrict__ q, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        double2 v[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const size_t k = i + j * stride;
            if (k < n) {
                if (NT_LD) { v[j].x = __builtin_nontemporal_load(&p[k].x); v[j].y = __builtin_nontemporal_load(&p[k].y); }
                else v[j] = p[k];
            }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const size_t k = i + j * stride;
            if (k < n) {
                if (NT_ST) { __builtin_nontemporal_store(v[j].x, &q[k].x); __builtin_nontemporal_store(v[j].y, &q[k].y); }
                else q[k] = v[j];
            }
        }
    }
}
// the solver's shape: one wavefront per workgroup walks its own slab; reads rows of 512 B, writes rows of 512 B
template <int DEPTH, bool NT>
__global__ __launch_bounds__(64) void k_tile_copy(const double* __restrict__ p, double* __restrict__ q, size_t rows_per_tile) {
    const double* src = p + (size_t)blockIdx.x * rows_per_tile * 64 + threadIdx.x;
    double* dst = q + (size_t)blockIdx.x * rows_per_tile * 64 + threadIdx.x;
    double ring[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; i++) ring[i] = NT ? __builtin_nontemporal_load(&src[(size_t)i * 64]) : src[(size_t)i * 64];
    for (size_t r = 0; r < rows_per_tile; r += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            const double v = ring[i];
            const size_t rn = r + i + DEPTH < rows_per_tile ? r + i + DEPTH : rows_per_tile - 1;
            ring[i] = NT ? __builtin_nontemporal_load(&src[rn * 64]) : src[rn * 64];
            if (NT) __builtin_nontemporal_store(v, &dst[(r + i) * 64]); else dst[(r + i) * 64] = v;
        }
    }
}
template <typename F>
static double timeit(F f, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main() {
    const size_t bytes = 4ull << 30, n2 = bytes / 16;
    double2 *p, *q;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&q, bytes) != hipSuccess) return 1;
    (void)hipMemset(p, 0, bytes); (void)hipMemset(q, 0, bytes);
#define RUN(U, L, S, G) printf("copy  unroll %d  nt-load %d  nt-store %d  grid %5d: %.2f TB/s (read+write bytes)\n", U, L, S, G, \
        2.0 * bytes / timeit([&] { hipLaunchKernelGGL((k_copy<U, L, S>), dim3(G), dim3(256), 0, 0, p, q, n2); }, 5) / 1e9)
    for (int g : {2048, 8192}) {
        RUN(1, false, false, g); RUN(4, false, false, g); RUN(8, false, false, g);
        RUN(4, true, true, g); RUN(8, true, true, g); RUN(4, false, true, g); RUN(4, true, false, g);
    }
    for (int tiles : {2048, 4096}) {
        const size_t rows = bytes / 512 / tiles;
#define RUNT(D, NT) printf("tile copy, %4d wavefronts, depth %2d, nt %d: %.2f TB/s\n", tiles, D, NT, \
        2.0 * bytes / timeit([&] { hipLaunchKernelGGL((k_tile_copy<D, NT>), dim3(tiles), dim3(64), 0, 0, (const double*)p, (double*)q, rows); }, 3) / 1e9)
        RUNT(8, false); RUNT(8, true); RUNT(32, false); RUNT(32, true);
    }
    // hipMemcpy device-to-device for reference
    printf("hipMemcpyDtoD: %.2f TB/s\n", 2.0 * bytes / timeit([&] { (void)hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, 0); }, 5) / 1e9);
    return 0;
}
