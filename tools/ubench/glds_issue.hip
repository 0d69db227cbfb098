// micro-benchmark: issue cost of LDS-DMA vs plain loads vs stores, 1 or 15 active waves (one workgroup of 1024)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned long long now() {
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
template <int MODE>
__global__ __launch_bounds__(1024) void k(float4 *pts, unsigned *rank, int n, int active, unsigned long long *out, float *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slots[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= active) return;
    unsigned char *my = slots + wave * 5120;
    unsigned long long c_issue = 0, c_wait = 0;
    float acc = 0.f;
    unsigned seed = wave * 7919u + 13u;
    for (int it = 0; it < 2000; it++) {
        seed = seed * 1664525u + 1013904223u;
        const int b = __builtin_amdgcn_readfirstlane((seed >> 8) % (n / 64));
        const int pos = b * 64 + lane;
        unsigned long long t0 = now();
        if (MODE == 0) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pts + pos),
                                             (__attribute__((address_space(3))) void *)(my), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rank + pos),
                                             (__attribute__((address_space(3))) void *)(my + 1024), 4, 0, 0);
            unsigned long long t1 = now();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            unsigned long long t2 = now();
            c_issue += t1 - t0; c_wait += t2 - t1;
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 pv;
            const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) void *)(my + lane * 16);
            asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pv) : "v"(a) : "memory");
            acc += pv.x + pv.w;
        } else if (MODE == 1) {
            float4 p = pts[pos];
            unsigned r = rank[pos];
            unsigned long long t1 = now();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            acc += p.x + p.w + r;
            unsigned long long t2 = now();
            c_issue += t1 - t0; c_wait += t2 - t1;
        } else if (MODE == 2) {  // store + glds: does an outstanding store slow the DMA issue?
            reinterpret_cast<float *>(pts + pos)[3] = acc + it;
            unsigned long long t1 = now();
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pts + pos),
                                             (__attribute__((address_space(3))) void *)(my), 16, 0, 0);
            unsigned long long t2 = now();
            c_issue += t1 - t0; c_wait += t2 - t1;
            __builtin_amdgcn_s_waitcnt(0x0F70);
        } else if (MODE == 4) {  // strided 4-byte store (the .w of 64 float4) -> time until it is acknowledged
            reinterpret_cast<float *>(pts + pos)[3] = acc + it;
            unsigned long long t1 = now();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            unsigned long long t2 = now();
            c_issue += t1 - t0; c_wait += t2 - t1;
        } else if (MODE == 5) {  // contiguous 4-byte store (a separate distance array)
            reinterpret_cast<float *>(rank)[pos] = acc + it;
            unsigned long long t1 = now();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            unsigned long long t2 = now();
            c_issue += t1 - t0; c_wait += t2 - t1;
        } else if (MODE == 6) {  // strided store, then DMA of ANOTHER bucket, wait for both (what an update does)
            reinterpret_cast<float *>(pts + pos)[3] = acc + it;
            const int pos2 = ((b * 7 + 13) % (n / 64)) * 64 + lane;
            unsigned long long t1 = now();
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pts + pos2),
                                             (__attribute__((address_space(3))) void *)(my), 16, 0, 0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            unsigned long long t2 = now();
            c_issue += t1 - t0; c_wait += t2 - t1;
        } else if (MODE == 3) {  // 4 DMA back to back
            for (int u = 0; u < 4; u++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pts + ((pos + u * 6400) % n)),
                                                 (__attribute__((address_space(3))) void *)(my + u * 1024), 16, 0, 0);
            unsigned long long t1 = now();
            __builtin_amdgcn_s_waitcnt(0x0F70);
            unsigned long long t2 = now();
            c_issue += t1 - t0; c_wait += t2 - t1;
        }
    }
    if (lane == 0) { out[wave * 2] = c_issue; out[wave * 2 + 1] = c_wait; }
    if (acc == 123.456f) *sink = acc;
}
int main() {
    const int n = 100032;
    float4 *pts; unsigned *rank; unsigned long long *out; float *sink;
    hipMalloc(&pts, n * 16); hipMalloc(&rank, n * 4); hipMalloc(&out, 32 * 8); hipMalloc(&sink, 4);
    hipMemset(pts, 0, n * 16); hipMemset(rank, 0, n * 4);
    unsigned long long h[32];
#define RUN(M, ACT)                                                                                            \
    hipFuncSetAttribute((const void *)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 100000);               \
    for (int rep = 0; rep < 2; rep++) {                                                                        \
        hipLaunchKernelGGL(k<M>, dim3(1), dim3(1024), 90000, 0, pts, rank, n, ACT, out, sink);                 \
        hipDeviceSynchronize();                                                                                \
    }                                                                                                          \
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);                                                       \
    printf("mode %d active %2d: wave0 first %.0f second %.0f | last wave first %.0f second %.0f (cycles/iter)\n", M, ACT, h[0] / 2000.0, h[1] / 2000.0, \
           h[(ACT - 1) * 2] / 2000.0, h[(ACT - 1) * 2 + 1] / 2000.0);
    RUN(0, 1) RUN(0, 4) RUN(0, 15) RUN(1, 1) RUN(1, 4) RUN(1, 15) RUN(2, 1) RUN(2, 15) RUN(4, 1) RUN(4, 15) RUN(5, 1) RUN(5, 15) RUN(6, 1) RUN(6, 15)
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
