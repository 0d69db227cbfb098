// I1: furthest point sampling, gfx950.
//
// Replaces lib/pointops2/src/sampling/sampling_cuda_kernel.cu:14-171 behind the same launcher.
//
// Bit-exactness.  The reference picks, per iteration, argmax_k min-dist[k] with two tie rules that
// depend on its launch geometry (block size B = opt_n_threads(n), cuda_utils.h:10-13):
//   - inside a thread (points k = start+tid, start+tid+B, ...) the FIRST maximum wins (strict '>',
//     sampling_cuda_kernel.cu:57-58);
//   - across threads the LDS tree (:64-123) lets slot t absorb slot t+s for s = B/2 ... 1 and keep
//     its own entry on ties, i.e. the thread with the smallest BIT-REVERSED id wins.
// Both rules are folded into one 64-bit key  (d2 bits << 32) | (0x7fffffff - (bitrev(k mod B) << 21 | k div B))
// whose plain unsigned maximum is the reference's winner, so any reduction shape — and any launch
// geometry — reproduces the reference's index sequence.  Squared distances use the same fma
// contraction as the oracle: fma(dz,dz, fma(dx,dx, dy*dy)).
//
// This file holds the single-workgroup-per-batch-element kernel (whole cloud streamed from L2 each
// iteration).  It is the baseline and the small-n path.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace p2 {

__device__ __forceinline__ float sqdist(float x1, float y1, float z1, float x2, float y2, float z2) {
    const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
    return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned long long o = __shfl_xor(v, s, 64);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ unsigned long long fps_key(float d2, int rel, int Bref, int log2B) {
    const unsigned tref = (unsigned)rel & (unsigned)(Bref - 1);
    const unsigned cidx = (unsigned)rel >> log2B;
    const unsigned brev = log2B ? (__brev(tref) >> (32 - log2B)) : 0u;
    const unsigned key2 = (brev << 21) | cidx;
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(0x7fffffffu - key2);
}
__device__ __forceinline__ int fps_decode(unsigned long long key, int Bref, int log2B) {
    const unsigned key2 = 0x7fffffffu - (unsigned)(key & 0xffffffffull);
    const unsigned brev = key2 >> 21, cidx = key2 & ((1u << 21) - 1);
    const unsigned tref = log2B ? (__brev(brev) >> (32 - log2B)) : 0u;
    return (int)(cidx * (unsigned)Bref + tref);
}

template <int BS>
__global__ __launch_bounds__(BS) void fps_block_kernel(int Bref, int log2B, const float *__restrict__ xyz,
                                                       const int *__restrict__ offset,
                                                       const int *__restrict__ new_offset, float *__restrict__ tmp,
                                                       int *__restrict__ idx) {
    __shared__ unsigned long long red[BS / 64];
    __shared__ unsigned long long winner;
    const int bid = blockIdx.x, tid = threadIdx.x;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    if (end_n <= start_n) {
        for (int j = start_m + tid; j < end_m; j += BS) idx[j] = start_n;
        return;
    }
    if (tid == 0 && start_m < end_m) idx[start_m] = start_n;
    int old = start_n;
    for (int j = start_m + 1; j < end_m; j++) {
        const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
        unsigned long long best = 0ull;
        for (int k = start_n + tid; k < end_n; k += BS) {
            const float d = sqdist(x1, y1, z1, xyz[k * 3 + 0], xyz[k * 3 + 1], xyz[k * 3 + 2]);
            const float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            const unsigned long long key = fps_key(d2, k - start_n, Bref, log2B);
            best = key > best ? key : best;
        }
        best = wave_max_u64(best);
        if ((tid & 63) == 0) red[tid >> 6] = best;
        __syncthreads();
        if (tid < 64) {
            unsigned long long v = tid < BS / 64 ? red[tid] : 0ull;
            v = wave_max_u64(v);
            if (tid == 0) winner = v;
        }
        __syncthreads();
        old = start_n + fps_decode(winner, Bref, log2B);
        if (tid == 0) idx[j] = old;
    }
}

// cuda_utils.h:10-13, same double-precision log quotient as the reference's host code (it decides
// the tie rule, so it is restated literally rather than with an integer log2)
static int ref_block_size(int n) {
    if (n < 1) return 1;
    const int pow_2 = (int)(std::log(static_cast<double>(n)) / std::log(2.0));
    return std::max(std::min(1 << pow_2, 1024), 1);
}

}  // namespace p2

using namespace p2;

extern "C" {

void furthestsampling_cuda_launcher(int b, int n, const float *xyz, const int *offset,
                                    const int *new_offset, float *tmp, int *idx) {
    if (b <= 0) return;
    const int Bref = ref_block_size(n);
    int log2B = 0;
    while ((1 << log2B) < Bref) log2B++;
    hipStream_t st = state().stream;
    const int N_total = state().total_points;
    state().total_points = 0;
    // bucketed exact FPS (fps_bucket.hip) needs a caller-provided workspace and the total point count
    if (n >= 2048 && fps_bucket_launch(b, n, Bref, log2B, xyz, offset, new_offset, N_total, idx)) {
        check_launch();
        return;
    }
    if (n > 4096)
        hipLaunchKernelGGL(fps_block_kernel<1024>, dim3(b), dim3(1024), 0, st, Bref, log2B, xyz, offset, new_offset, tmp, idx);
    else
        hipLaunchKernelGGL(fps_block_kernel<256>, dim3(b), dim3(256), 0, st, Bref, log2B, xyz, offset, new_offset, tmp, idx);
    check_launch();
}

}  // extern "C"
