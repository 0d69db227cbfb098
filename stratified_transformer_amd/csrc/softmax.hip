// A3: segment softmax over the CSR pair list, forward and backward, gfx950.
//
// The model calls torch_scatter.scatter_softmax(attn_flat [M,h], index_0, dim=0)
// (model/stratified_transformer.py:205; torch_scatter 2.0.6 composite/softmax.py: segment max,
// exp(x - max), segment sum, divide).  index_0 is ascending there, so segments are the CSR ranges.
//
// Mapping: one wavefront per query.  A pair's h head values sit in HP = next_pow2(h) adjacent lanes,
// so a wave covers 64/HP pairs per pass, global accesses are one contiguous run, and the per-head
// reductions are xor-butterflies over the pair-slot bits of the lane id.  A segment (<= 1024 pairs)
// is at most a few KB, so the three passes (max, exp+sum, normalise) re-read it from L1.
#include "common.h"

namespace p2 {

__device__ __forceinline__ float slot_max(float v, int hp) {
    for (int s = hp; s < 64; s <<= 1) v = fmaxf(v, __shfl_xor(v, s, 64));
    return v;
}
__device__ __forceinline__ float slot_sum(float v, int hp) {
    for (int s = hp; s < 64; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}

__global__ __launch_bounds__(256) void seg_softmax_fwd_kernel(int N, int M, int h, int hp, const float *__restrict__ src,
                                                              const int *__restrict__ offs, float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= N) return;
    const int ppw = 64 / hp, p = lane / hp, c = lane % hp;
    const int s = max(offs[qi], 0), e = min(offs[qi + 1], M);  // never outside [0, M), whatever the offsets hold
    if (e <= s) return;
    for (int hb = 0; hb < h; hb += hp) {
        const int hh = hb + c;
        const bool hv = hh < h;
        float mx = -INFINITY;
        for (int m = s + p; m < e; m += ppw)
            if (hv) mx = fmaxf(mx, src[(size_t)m * h + hh]);
        mx = slot_max(mx, hp);
        float sum = 0.f;
        for (int m = s + p; m < e; m += ppw)
            if (hv) {
                const float ex = expf(src[(size_t)m * h + hh] - mx);
                out[(size_t)m * h + hh] = ex;
                sum += ex;
            }
        sum = slot_sum(sum, hp);
        for (int m = s + p; m < e; m += ppw)
            if (hv) out[(size_t)m * h + hh] = out[(size_t)m * h + hh] / sum;
    }
}

// grad_src = y * (grad_y - sum_seg(y * grad_y))
__global__ __launch_bounds__(256) void seg_softmax_bwd_kernel(int N, int M, int h, int hp, const float *__restrict__ y,
                                                              const float *__restrict__ gy, const int *__restrict__ offs,
                                                              float *__restrict__ gx) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + wave;
    if (qi >= N) return;
    const int ppw = 64 / hp, p = lane / hp, c = lane % hp;
    const int s = max(offs[qi], 0), e = min(offs[qi + 1], M);  // never outside [0, M), whatever the offsets hold
    if (e <= s) return;
    for (int hb = 0; hb < h; hb += hp) {
        const int hh = hb + c;
        const bool hv = hh < h;
        float dot = 0.f;
        for (int m = s + p; m < e; m += ppw)
            if (hv) dot = fmaf(y[(size_t)m * h + hh], gy[(size_t)m * h + hh], dot);
        dot = slot_sum(dot, hp);
        for (int m = s + p; m < e; m += ppw)
            if (hv) gx[(size_t)m * h + hh] = y[(size_t)m * h + hh] * (gy[(size_t)m * h + hh] - dot);
    }
}

// Few rows, many heads (the late stages: h = 12 / 24 gives only 4 / 2 pairs per wave pass): one WORKGROUP per
// query, its four waves stride over the pairs together and combine their partial max / sum through LDS.
// Same arithmetic per element; the sum's order differs (fp32 tolerance, like the wave version's butterfly).
__device__ __forceinline__ float block_combine(float v, int hp, bool is_max, float *red, int wave, int lane) {
    v = is_max ? slot_max(v, hp) : slot_sum(v, hp);
    __syncthreads();  // red[] from the previous use has been read
    if (lane < hp) red[wave * 64 + lane] = v;
    __syncthreads();
    float r = red[lane % hp];
#pragma unroll
    for (int w = 1; w < 4; w++) {
        const float o = red[w * 64 + lane % hp];
        r = is_max ? fmaxf(r, o) : r + o;
    }
    return r;
}

__global__ __launch_bounds__(256) void seg_softmax_fwd_block_kernel(int N, int M, int h, int hp, const float *__restrict__ src,
                                                                    const int *__restrict__ offs, float *__restrict__ out) {
    __shared__ float red[4 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x;
    const int ppw = 64 / hp, p = lane / hp + wave * ppw, c = lane % hp, stride = 4 * ppw;
    const int s = max(offs[qi], 0), e = min(offs[qi + 1], M);  // never outside [0, M), whatever the offsets hold
    if (e <= s) return;
    for (int hb = 0; hb < h; hb += hp) {
        const int hh = hb + c;
        const bool hv = hh < h;
        float mx = -INFINITY;
        for (int m = s + p; m < e; m += stride)
            if (hv) mx = fmaxf(mx, src[(size_t)m * h + hh]);
        mx = block_combine(mx, hp, true, red, wave, lane);
        float sum = 0.f;
        for (int m = s + p; m < e; m += stride)
            if (hv) {
                const float ex = expf(src[(size_t)m * h + hh] - mx);
                out[(size_t)m * h + hh] = ex;
                sum += ex;
            }
        sum = block_combine(sum, hp, false, red, wave, lane);
        for (int m = s + p; m < e; m += stride)
            if (hv) out[(size_t)m * h + hh] = out[(size_t)m * h + hh] / sum;
    }
}

__global__ __launch_bounds__(256) void seg_softmax_bwd_block_kernel(int N, int M, int h, int hp, const float *__restrict__ y,
                                                                    const float *__restrict__ gy, const int *__restrict__ offs,
                                                                    float *__restrict__ gx) {
    __shared__ float red[4 * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int qi = blockIdx.x;
    const int ppw = 64 / hp, p = lane / hp + wave * ppw, c = lane % hp, stride = 4 * ppw;
    const int s = max(offs[qi], 0), e = min(offs[qi + 1], M);  // never outside [0, M), whatever the offsets hold
    if (e <= s) return;
    for (int hb = 0; hb < h; hb += hp) {
        const int hh = hb + c;
        const bool hv = hh < h;
        float dot = 0.f;
        for (int m = s + p; m < e; m += stride)
            if (hv) dot = fmaf(y[(size_t)m * h + hh], gy[(size_t)m * h + hh], dot);
        dot = block_combine(dot, hp, false, red, wave, lane);
        for (int m = s + p; m < e; m += stride)
            if (hv) gx[(size_t)m * h + hh] = y[(size_t)m * h + hh] * (gy[(size_t)m * h + hh] - dot);
    }
}

static bool few_rows_many_heads(int N, int h) { return N < 20000 && h > 4; }

static int next_pow2_le64(int h) {
    int hp = 1;
    while (hp < h && hp < 64) hp <<= 1;
    return hp;
}

}  // namespace p2

using namespace p2;

extern "C" {

void segment_softmax_forward_launcher(int N, int M, int h, const float *src, const int *offsets, float *out) {
    if (N <= 0 || M <= 0) return;
    if (few_rows_many_heads(N, h))
        hipLaunchKernelGGL(seg_softmax_fwd_block_kernel, dim3(N), dim3(256), 0, state().stream, N, M, h, next_pow2_le64(h), src, offsets, out);
    else
        hipLaunchKernelGGL(seg_softmax_fwd_kernel, dim3(div_up(N, 4)), dim3(256), 0, state().stream, N, M, h, next_pow2_le64(h), src, offsets, out);
    check_launch();
}

void segment_softmax_backward_launcher(int N, int M, int h, const float *out, const float *grad_out,
                                       const int *offsets, float *grad_src) {
    if (N <= 0 || M <= 0) return;
    if (few_rows_many_heads(N, h))
        hipLaunchKernelGGL(seg_softmax_bwd_block_kernel, dim3(N), dim3(256), 0, state().stream, N, M, h, next_pow2_le64(h), out, grad_out, offsets, grad_src);
    else
        hipLaunchKernelGGL(seg_softmax_bwd_kernel, dim3(div_up(N, 4)), dim3(256), 0, state().stream, N, M, h, next_pow2_le64(h), out, grad_out, offsets, grad_src);
    check_launch();
}

}  // extern "C"
