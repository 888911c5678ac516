// Probe: does a wavefront with fewer active lanes issue VALU instructions faster on gfx950, and what does a lone
// wavefront pay per instruction with 1, 2, 4 or 8 independent fp64 chains?  One wavefront per workgroup, 256 workgroups
// (one per CU), s_memtime around the loop; prints core clocks per instruction.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/exec_probe tools/probes/exec_mask_probe.hip && /tmp/exec_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N_IT = 2048;
template <int CH>
__global__ __launch_bounds__(64) void k(unsigned long long* out, double seed, int lanes) {
    double a[8];
#pragma unroll
    for (int j = 0; j < 8; j++) a[j] = seed + j * 0.125 + threadIdx.x * 1e-3;
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    unsigned long long t0 = 0, t1 = 0;
    if ((int)threadIdx.x < lanes) {   // EXEC = the low `lanes` lanes for the whole loop
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < N_IT; it++) {
#pragma unroll
            for (int r = 0; r < 8 / CH; r++)
#pragma unroll
                for (int j = 0; j < CH; j++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c1), "v"(c2));
        }
        t1 = __builtin_amdgcn_s_memtime();
    }
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += a[j];
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = (unsigned long long)s; }
}
template <int CH>
static double run(unsigned long long* d, int lanes) {
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(64), 0, 0, d, 1.0001, lanes);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(512);
    (void)hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; i++) s += (double)h[2 * i];
    return s / 256 / (N_IT * 8.0);
}
int main() {
    unsigned long long* d;
    (void)hipMalloc(&d, 4096);
    run<8>(d, 64);
    printf("clocks per v_fma_f64 of a lone wavefront (s_memtime ticks)\n");
    printf("%-28s %8s %8s %8s %8s\n", "active lanes", "1 chain", "2 chains", "4 chains", "8 chains");
    for (int lanes : {64, 32, 16, 8, 1})
        printf("%-28d %8.2f %8.2f %8.2f %8.2f\n", lanes, run<1>(d, lanes), run<2>(d, lanes), run<4>(d, lanes), run<8>(d, lanes));
    return 0;
}
