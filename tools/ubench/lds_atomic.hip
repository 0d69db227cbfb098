// micro-benchmark: throughput of LDS float atomic adds (ds_add_f32) vs plain LDS read-modify-write,
// for conflict-free, random-bin and same-address patterns; 1 wave, 4 waves (one per SIMD), 16 waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(1024) void k(int pattern, int active_lanes, unsigned long long *out, float *sink) {
    __shared__ float hist[16][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lane; i < 1024; i += 64) hist[wave][i] = 0.f;
    __syncthreads();
    unsigned seed = threadIdx.x * 2654435761u + 12345u;
    const int iters = 4096;
    int bins[8];
    for (int u = 0; u < 8; u++) {
        seed = seed * 1664525u + 1013904223u;
        if (pattern == 0) bins[u] = (lane + u * 64) & 1023;            // conflict-free, distinct
        else if (pattern == 1) bins[u] = (seed >> 10) & 63;              // random among 64 bins (2 per bank)
        else if (pattern == 2) bins[u] = ((seed >> 10) & 63) + (lane & 3) * 65;  // random, 4 planes with padded stride
        else bins[u] = u;                                               // all lanes the same address
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float w = 1.0f + lane;
    if (lane < active_lanes) {
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                if (MODE == 0) atomicAdd(&hist[wave][bins[u]], w);
                else if (MODE == 1) hist[wave][bins[u]] += w;   // racy plain RMW: cost reference only
                else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned *>(&hist[wave][bins[u]]), (unsigned)lane);  // ds_add_u32
                else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long *>(&hist[wave][bins[u] & ~1]), (unsigned long long)lane);  // ds_add_u64
                else hist[wave][bins[u]] = w;                    // plain write
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
    __syncthreads();
    if (hist[wave][lane] == 123.456f) *sink = 1.f;
}
int main() {
    unsigned long long *out; float *sink;
    hipMalloc(&out, 16 * 8 * 512); hipMalloc(&sink, 4);
    unsigned long long h[16];
    const char *pn[] = {"distinct", "random64", "random64 padded planes", "same address"};
    const char *mn[] = {"ds_add_f32", "plain rmw", "plain write", "ds_add_u32", "ds_add_u64"};
#define RUN(M, P, WAVES, LANES)                                                                     \
    hipLaunchKernelGGL(k<M>, dim3(1), dim3(WAVES * 64), 0, 0, P, LANES, out, sink);                  \
    hipDeviceSynchronize();                                                                          \
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);                                             \
    printf("%-12s %-24s waves %2d lanes %2d: %.1f cycles per wave-instruction (wave 0)\n", mn[M], pn[P], WAVES, LANES, h[0] / 4096.0);
    for (int p = 0; p < 4; p++) {
        RUN(0, p, 1, 64) RUN(0, p, 4, 64) RUN(0, p, 16, 64) RUN(0, p, 16, 48)
        RUN(1, p, 16, 64) RUN(2, p, 16, 64) RUN(3, p, 1, 64) RUN(3, p, 16, 64) RUN(4, p, 1, 64) RUN(4, p, 16, 64)
    }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
