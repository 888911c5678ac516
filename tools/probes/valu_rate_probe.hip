// Probe: issue cost of the VALU instructions the rollout kernels are made of, relative to v_fma_f64.
// One wavefront per SIMD (grid = 1024 workgroups of 64), 8 independent chains per lane so that the
// dependent-issue latency is hidden, N_IT iterations of 8 x UNROLL instructions each.  Reports ns per
// instruction per wavefront and the ratio to fma; with W wavefronts per SIMD (second table) it shows
// whether two wavefronts overlap on one SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_probe tools/probes/valu_rate_probe.hip && /tmp/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int N_IT = 4096;

#define LOOP8(BODY)                                   \
    for (int it = 0; it < N_IT; it++) {               \
        _Pragma("unroll") for (int j = 0; j < 8; j++) { BODY; } \
    }

template <int OP>
__global__ __launch_bounds__(64) void k(double* out, double seed, int iseed) {
    double a[8];
    float f[8];
    int q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { a[j] = seed + j * 0.125 + threadIdx.x * 1e-3; f[j] = (float)a[j]; q[j] = iseed + j; }
    const double c1 = seed * 0.999, c2 = seed * 1e-3;
    if (OP == 0) LOOP8(a[j] = __builtin_fma(a[j], c1, c2))
    if (OP == 1) LOOP8(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[j]) : "v"(a[j])); asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[j]) : "v"(f[j])))
    if (OP == 2) LOOP8(asm volatile("v_rcp_f64 %0, %1" : "=v"(a[j]) : "v"(a[j])))
    if (OP == 3) LOOP8(asm volatile("v_rndne_f64 %0, %1" : "=v"(a[j]) : "v"(a[j])))
    if (OP == 4) LOOP8(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(q[j]) : "v"(a[j])); asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[j]) : "v"(q[j])))
    if (OP == 5) LOOP8(asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(q[j]) : "v"(q[j]), "v"(iseed)))
    if (OP == 6) LOOP8(a[j] = c1 / a[j])   // full IEEE division sequence
    if (OP == 7) LOOP8(asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[j]) : "v"(a[j]), "v"(c1)))
    if (OP == 8) LOOP8(asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[j]) : "v"(a[j]), "v"(c2)))
    if (OP == 9) LOOP8(asm volatile("v_xor_b32 %0, %1, %2" : "=v"(q[j]) : "v"(q[j]), "v"(iseed)))
    if (OP == 10) LOOP8(asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f[j]) : "v"(f[j]), "v"((float)c1), "v"((float)c2)))
    if (OP == 11) LOOP8(asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a[j]), "v"(c1) : "vcc"))
    if (OP == 12) LOOP8(asm volatile("v_mov_b32 %0, %1" : "=v"(q[j]) : "v"(q[j])))
    if (OP == 13) LOOP8(asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(a[j]) : "v"(a[j]), "v"(c1), "v"(c2)))
    if (OP == 14) LOOP8(asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(q[j]) : "v"(q[j]), "v"(iseed), "s"((unsigned long long)iseed * 0x9e3779b97f4a7c15ull)))
    if (OP == 15) LOOP8(asm volatile("v_bfi_b32 %0, %1, %2, %3" : "=v"(q[j]) : "v"(iseed), "v"(q[j]), "v"(iseed + 7)))
    if (OP == 16) LOOP8(asm volatile("v_div_fmas_f64 %0, %1, %2, %3" : "=v"(a[j]) : "v"(a[j]), "v"(c1), "v"(c2)))
    if (OP == 17) LOOP8(asm volatile("v_div_fixup_f64 %0, %1, %2, %3" : "=v"(a[j]) : "v"(a[j]), "v"(c1), "v"(c2)))
    if (OP == 18) LOOP8(asm volatile("v_div_scale_f64 %0, vcc, %1, %2, %3" : "=v"(a[j]) : "v"(a[j]), "v"(c1), "v"(a[j]) : "vcc"))
    if (OP == 19) LOOP8(int sq; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sq) : "v"(q[j])); asm volatile("v_add_u32 %0, %1, %2" : "=v"(q[j]) : "s"(sq), "v"(q[j])))
    if (OP == 20) LOOP8(asm volatile("v_cmp_ne_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %1, %2, vcc" : "=v"(q[j]) : "v"(q[j]), "v"(iseed) : "vcc"))
    if (OP == 21) LOOP8(asm volatile("v_max_f64 %0, %1, %2" : "=v"(a[j]) : "v"(a[j]), "v"(c1)))
    double s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) s += a[j] + f[j] + q[j];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP>
static float run(int grid, double* d) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, 1.0001, 3);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, d, 1.0001, 3);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    double* d;
    (void)hipMalloc(&d, 8192 * 64 * 8);
    const char* names[] = {"v_fma_f64", "cvt f64->f32->f64 (2)", "v_rcp_f64", "v_rndne_f64", "cvt f64->i32->f64 (2)",
                           "v_cndmask_b32", "IEEE f64 division", "v_mul_f64", "v_add_f64", "v_xor_b32", "v_fma_f32",
                           "v_cmp_gt_f64", "v_mov_b32", "v_pk_fma_f32", "v_cndmask_b32_e64 sgpr",
                           "v_bfi_b32", "v_div_fmas_f64", "v_div_fixup_f64", "v_div_scale_f64", "readlane+add_u32 (2)",
                           "v_cmp_u32 + cndmask (2)", "v_max_f64"};
    for (int grid : {1024, 2048}) {
        float t[22];
        t[0] = run<0>(grid, d); t[1] = run<1>(grid, d); t[2] = run<2>(grid, d); t[3] = run<3>(grid, d);
        t[4] = run<4>(grid, d); t[5] = run<5>(grid, d); t[6] = run<6>(grid, d); t[7] = run<7>(grid, d);
        t[8] = run<8>(grid, d); t[9] = run<9>(grid, d); t[10] = run<10>(grid, d); t[11] = run<11>(grid, d);
        t[12] = run<12>(grid, d); t[13] = run<13>(grid, d);
        t[14] = run<14>(grid, d); t[15] = run<15>(grid, d); t[16] = run<16>(grid, d); t[17] = run<17>(grid, d);
        t[18] = run<18>(grid, d); t[19] = run<19>(grid, d); t[20] = run<20>(grid, d); t[21] = run<21>(grid, d);
        printf("grid %d wavefronts (%d per SIMD)\n", grid, grid / 1024);
        for (int i = 0; i < 22; i++) {
            const double per = t[i] * 1e6 / (N_IT * 8.0);  // ns per loop-body statement per wavefront
            printf("  %-26s %8.3f ms  %7.2f ns/stmt  x%.2f of fma\n", names[i], t[i], per, t[i] / t[0]);
        }
    }
    return 0;
}
