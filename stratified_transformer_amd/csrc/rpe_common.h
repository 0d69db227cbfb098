// Shared device helpers of the rel-pos kernels (rpe.hip, rpe_bwd_mfma.hip): LDS table images and the
// wave-per-row walker prologue.
#pragma once
#include "common.h"

namespace p2 {

constexpr size_t kLdsBudget = 72 * 1024;  // two workgroups per CU

// LDS image of one table slice: [hg][3][L][D]
template <int D>
__device__ __forceinline__ void stage_table(float *lds, const float *__restrict__ tab, int L, int h, int h0, int hgn) {
    const int total = hgn * 3 * L * D;
    for (int x = threadIdx.x; x < total; x += blockDim.x) {
        const int i = x % D;
        const int r = (x / D) % L;
        const int ax = (x / (D * L)) % 3;
        const int t = x / (D * L * 3);
        lds[x] = tab[(((size_t)r * h + (h0 + t)) * D + i) * 3 + ax];
    }
}
template <int D>
__device__ __forceinline__ void zero_lds(float *lds, int n) {
    for (int x = threadIdx.x; x < n; x += blockDim.x) lds[x] = 0.f;
}
// adds the LDS gradient image back into the global [L,h,D,3] table
template <int D>
__device__ __forceinline__ void flush_table(const float *lds, float *__restrict__ gtab, int L, int h, int h0, int hgn) {
    const int total = hgn * 3 * L * D;
    for (int x = threadIdx.x; x < total; x += blockDim.x) {
        const float v = lds[x];
        if (v != 0.f) {
            const int i = x % D;
            const int r = (x / D) % L;
            const int ax = (x / (D * L)) % 3;
            const int t = x / (D * L * 3);
            atomicAdd(gtab + (((size_t)r * h + (h0 + t)) * D + i) * 3 + ax, v);
        }
    }
}

template <int D>
__device__ __forceinline__ const float4 *trow(const float *lds, int L, int t, int ax, int r, int c) {
    return reinterpret_cast<const float4 *>(lds + (((size_t)t * 3 + ax) * L + r) * D + 4 * c);
}
// T(m, head t)[4c..4c+3] = tab[r0,.,.,0] + tab[r1,.,.,1] + tab[r2,.,.,2]   (left to right, as the reference)
template <int D>
__device__ __forceinline__ float4 tsum(const float *lds, int L, int t, int r0, int r1, int r2, int c) {
    return add4(add4(*trow<D>(lds, L, t, 0, r0, c), *trow<D>(lds, L, t, 1, r1, c)), *trow<D>(lds, L, t, 2, r2, c));
}
template <int D>
__device__ __forceinline__ void tadd(float *lds, int L, int t, int r0, int r1, int r2, int c, float4 v) {
    float *a0 = lds + (((size_t)t * 3 + 0) * L + r0) * D + 4 * c;
    float *a1 = lds + (((size_t)t * 3 + 1) * L + r1) * D + 4 * c;
    float *a2 = lds + (((size_t)t * 3 + 2) * L + r2) * D + 4 * c;
    atomicAdd(a0 + 0, v.x); atomicAdd(a0 + 1, v.y); atomicAdd(a0 + 2, v.z); atomicAdd(a0 + 3, v.w);
    atomicAdd(a1 + 0, v.x); atomicAdd(a1 + 1, v.y); atomicAdd(a1 + 2, v.z); atomicAdd(a1 + 3, v.w);
    atomicAdd(a2 + 0, v.x); atomicAdd(a2 + 1, v.y); atomicAdd(a2 + 2, v.z); atomicAdd(a2 + 3, v.w);
}
__device__ __forceinline__ int clampr(int r, int L) { return min(max(r, 0), L - 1); }

#define P2_WALK_PROLOGUE                                                            \
    constexpr int LPG = Geo<D>::LPG, PPW = Geo<D>::PPW;                             \
    extern __shared__ float lds[];                                                  \
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                     \
    const int C = h * D;                                                            \
    const int p = lane / LPG, c = lane % LPG;                                       \
    const int h0 = blockIdx.y * HG;                                                 \
    const int hgn = min(HG, h - h0);                                                \
    const int tsz = hgn * 3 * L * D;                                                \
    (void)PPW; (void)p; (void)tsz; (void)C; (void)c; (void)wave;

// a pair's key id and its three (unclamped) table rows, as one unit the walkers request a pass ahead
struct PairIds {
    int j, q0, q1, q2;
};
__device__ __forceinline__ PairIds load_pair_ids(const int *__restrict__ idx1, const int *__restrict__ rel, int m) {
    PairIds r;
    r.j = idx1[m];
    r.q0 = rel[m * 3 + 0];
    r.q1 = rel[m * 3 + 1];
    r.q2 = rel[m * 3 + 2];
    return r;
}

// rpe_fallback.hip: the rel-pos operators without a table length (global-memory tables, the reference's atomics)
void a2_fwd_global(int N, int M, int h, int d, const float *q, const int *offs, const float *k, const int *idxk, const float *tq,
                   const float *tk, const int *rel, float *out);
void a2_bwd_global(int N, int M, int h, int d, const float *go, const float *q, const int *offs, const float *k, const int *idxk,
                   const float *tq, const float *tk, const int *rel, float *gq, float *gk, float *gtq, float *gtk);
void a4_fwd_global(int N, int M, int h, int d, const float *attn, const float *v, const int *offs, const int *idx1, const float *tv,
                   const int *rel, float *out);
void a4_bwd_global(int N, int M, int h, int d, const float *go, const int *offs, const int *idx1, const float *attn, const float *v,
                   const float *tv, const int *rel, float *ga, float *gv, float *gt);

// rpe_bwd_mfma.hip
bool a2_bwd_mfma(int N, int NK, int M, int h, int hdim, int L, const float *go, const float *q, const int *offs, const float *k,
                 const float *table_q, const float *table_k, const int *rel, const int *co, const int *cp,
                 float *grad_q, float *grad_k, float *gtq, float *gtk);
bool a4_bwd_mfma(int N, int h, int hdim, int L, const float *go, const int *offs, const int *idx1, const float *attn,
                 const float *v, const float *table, const int *rel, float *grad_attn, float *grad_table, ForkJoin &fj);

bool wattn_bwd(int N, int NK, int M, int h, int hdim, int L, const float *go, const float *q, const float *k, const float *v,
               const float *attn, const int *offs, const int *idx1, const float *table_q, const float *table_k,
               const float *table_v, const int *rel, const int *co, const int *cp, const int *cq, float *grad_logit,
               float *grad_q, float *grad_k, float *grad_v, float *gtq, float *gtk, float *gtv);

}  // namespace p2
