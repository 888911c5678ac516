// Probe: where do global_load_lds_dwordx3 / dwordx4 put each lane's bytes in LDS?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void g_void;
template <int SZ>
__global__ void probe(const unsigned* __restrict__ src, unsigned* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[512];
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    if constexpr (SZ == 12)
        __builtin_amdgcn_global_load_lds((g_void*)((const char*)src + lane * 12), (lds_void*)lds, 12, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((g_void*)((const char*)src + lane * 16), (lds_void*)lds, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(1024);
    for (int i = 0; i < 1024; i++) h[i] = i;
    unsigned *d, *o;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&o, 2048);
    (void)hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
    std::vector<unsigned> r(512);
    for (int sz : {12, 16}) {
        if (sz == 12) hipLaunchKernelGGL(probe<12>, dim3(1), dim3(64), 0, 0, d, o);
        else hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, d, o);
        (void)hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
        int linear = 1;
        for (int i = 0; i < 64 * sz / 4; i++) if (r[i] != (unsigned)i) linear = 0;
        printf("size %d: linear=%d  first 16 dwords:", sz, linear);
        for (int i = 0; i < 16; i++) printf(" %u", r[i]);
        printf(" ... [64..71]:");
        for (int i = 64; i < 72; i++) printf(" %u", r[i]);
        printf("\n");
    }
    return 0;
}
