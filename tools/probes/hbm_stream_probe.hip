// Probe: what plain streaming reaches on this box, to calibrate the "achievable" HBM rate the kernels are
// compared with.  read: sum of a 4 GiB buffer; write: fill; copy: read + write; tile_walk: the access shape
// of the solver's passes (one wavefront per workgroup walking its own contiguous slab with 8-byte loads per
// lane, `depth` rows of 512 B in flight).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_probe tools/probes/hbm_stream_probe.hip && /tmp/hbm_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void k_read(const double2* __restrict__ p, size_t n, double* out) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const double2 v = p[i]; s += v.x + v.y; }
    if (s == 123.456) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(double2* __restrict__ p, size_t n, double v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = double2{v, v};
}
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ p, double2* __restrict__ q, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
template <int DEPTH>
__global__ __launch_bounds__(64) void k_tile_walk(const double* __restrict__ p, size_t rows_per_tile, double* out) {
    const double* base = p + (size_t)blockIdx.x * rows_per_tile * 64 + threadIdx.x;
    double ring[DEPTH], s = 0;
#pragma unroll
    for (int i = 0; i < DEPTH; i++) ring[i] = base[(size_t)i * 64];
    for (size_t r = 0; r < rows_per_tile; r += DEPTH) {
#pragma unroll
        for (int i = 0; i < DEPTH; i++) {
            s += ring[i];
            const size_t rn = r + i + DEPTH < rows_per_tile ? r + i + DEPTH : rows_per_tile - 1;
            ring[i] = base[rn * 64];
        }
    }
    if (s == 123.456) out[0] = s;
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
    double* out;
    if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&q, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(p, 0, bytes); (void)hipMemset(q, 0, bytes);
    for (int grid : {2048, 8192, 32768}) {
        const double tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, p, n2, out); }, 5);
        const double tw = timeit([&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, p, n2, 1.0); }, 5);
        const double tc = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, p, q, n2); }, 5);
        printf("grid %5d x256: read %.2f TB/s  write %.2f TB/s  copy %.2f TB/s (read+write bytes)\n", grid, bytes / tr / 1e9,
               bytes / tw / 1e9, 2.0 * bytes / tc / 1e9);
    }
    for (int tiles : {1024, 2048, 4096, 8192}) {
        const size_t rows = bytes / 512 / tiles;
        const double t2 = timeit([&] { hipLaunchKernelGGL(k_tile_walk<2>, dim3(tiles), dim3(64), 0, 0, (const double*)p, rows, out); }, 3);
        const double t8 = timeit([&] { hipLaunchKernelGGL(k_tile_walk<8>, dim3(tiles), dim3(64), 0, 0, (const double*)p, rows, out); }, 3);
        const double t32 = timeit([&] { hipLaunchKernelGGL(k_tile_walk<32>, dim3(tiles), dim3(64), 0, 0, (const double*)p, rows, out); }, 3);
        printf("tile walk, %4d wavefronts: depth 2 %.2f TB/s  depth 8 %.2f TB/s  depth 32 %.2f TB/s\n", tiles, bytes / t2 / 1e9,
               bytes / t8 / 1e9, bytes / t32 / 1e9);
    }
    return 0;
}
