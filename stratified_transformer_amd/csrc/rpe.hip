// A2 (contextual relative-position bias) and A4 (AV with relative-position value), gfx950.
//
// Replaces lib/pointops2/src/rpe_v2/relative_pos_encoding_cuda_kernel_v2.cu:247-540 (v3 bias, v2
// AV) and lib/pointops2/src/rpe/relative_pos_encoding_cuda_kernel.cu (v1 forms) behind the same
// launchers.
//
// Design
//  * The [L,h,D,3] tables are tiny (12 KB per head at L=64, D=16) and hit by every pair, so each
//    workgroup stages the heads it works on in LDS, TRANSPOSED to [head][axis][row][D]: the D
//    floats of one (axis,row,head) become one contiguous 64/128-byte run and a lane fetches its
//    quarter with a single ds_read_b128 (the reference does 3*D scattered 4-byte global loads per
//    pair and table).  Workgroups are persistent (grid-stride over queries) so the staging cost is
//    paid once per CU, and the grid's second dimension walks head groups that fit the LDS budget.
//  * Same wave-per-query walker as attention.hip (LPG lanes per head vector, PPW pairs per pass).
//  * Backward: table gradients are accumulated in LDS (ds_add_f32) and flushed once per workgroup
//    with global atomics — the reference issues 6-8 global atomics per (pair, channel) into a
//    9216-float table.  Key-side gradients are produced by key through the CSC view
//    (pointops2_set_csc); without it the by-query kernel falls back to global atomics.
#include "rpe_common.h"
#include <cstdlib>

namespace p2 {

// ------------------------------------------------------------------------------------------------
// A2 forward: out[m,hh] = sum_i q[query,hh,i]*Tq(m,hh,i) + k[idx_k[m],hh,i]*Tk(m,hh,i)
// ------------------------------------------------------------------------------------------------
// The per-query loops are latency-bound (index -> key row -> LDS table rows is a dependent chain and a
// segment is only ~3 passes long): time scales with 1/(waves per CU) (A2 forward, stage 0: 674 / 371 / 227 us
// at 4 / 8 / 16 waves per CU).  So workgroups are as large as the register budget allows: the LDS table
// images are shared by all waves of a workgroup, two (A2: 72 KB) or four (A4: 36 KB) of them per CU.
template <int D, int HG>
__global__ __launch_bounds__(768, 6) void a2_fwd_kernel(int N, int h, int L, const float *__restrict__ q,
                                                        const int *__restrict__ offs, const float *__restrict__ k,
                                                        const int *__restrict__ idx_k, const float *__restrict__ table_q,
                                                        const float *__restrict__ table_k, const int *__restrict__ rel,
                                                        float *__restrict__ out, const int *__restrict__ rord) {
    P2_WALK_PROLOGUE
    float *Tq = lds, *Tk = lds + tsz;
    stage_table<D>(Tq, table_q, L, h, h0, hgn);
    stage_table<D>(Tk, table_k, L, h, h0, hgn);
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    // No per-head guards inside the walk: a head slot beyond the group's last head (only possible in the last head
    // group when h is not a multiple of HG) repeats that last head and its result is never stored.  With a branch
    // per head the compiler ends each head's block with a wait for its own key-row load, i.e. HG dependent memory
    // round trips per pass instead of one.
    for (RowSlots slots(rord, N, wpb, wave); slots.more(); slots.next()) {
        const int qi = slots.row();
        float4 q4[HG];
#pragma unroll
        for (int t = 0; t < HG; t++) q4[t] = ldg4(q + (size_t)qi * C + (h0 + min(t, hgn - 1)) * D + 4 * c);
        const int s = offs[qi], e = offs[qi + 1];
        // the ids of the next pass are requested together with the key rows of this one: one round trip per pass
        PairIds nx = load_pair_ids(idx_k, rel, max(0, min(s + p, e - 1)));
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int m = m0 + p;
            const bool valid = m < e;
            const PairIds cur = nx;
            nx = load_pair_ids(idx_k, rel, min(m + PPW, e - 1));
            float4 k4[HG];
#pragma unroll
            for (int t = 0; t < HG; t++) k4[t] = ldg4(k + (size_t)cur.j * C + (h0 + min(t, hgn - 1)) * D + 4 * c);
            __builtin_amdgcn_sched_barrier(0);
            const int r0 = clampr(cur.q0, L), r1 = clampr(cur.q1, L), r2 = clampr(cur.q2, L);
            float keep = 0.f;
#pragma unroll
            for (int t = 0; t < HG; t++) {
                const int te = min(t, hgn - 1);
                float part = dot4(q4[t], tsum<D>(Tq, L, te, r0, r1, r2, c)) + dot4(k4[t], tsum<D>(Tk, L, te, r0, r1, r2, c));
                float tot = xor_sum<1, LPG>(part);
                if (c == t) keep = tot;
            }
            if (valid && c < hgn) out[(size_t)m * h + h0 + c] = keep;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused logits + softmax (optional fast path; not part of the reference's operator set):
//   attn[m, hh] = softmax over the query's pairs of ( <q, k[j]> + <q, Tq(m)> + <k[j], Tk(m)> )
// = A1 + A2 + add + A3 in one walk: the key rows are gathered once instead of twice and three [M, h]
// round trips disappear.  Every term is computed exactly as the separate operators compute it (same dot4 /
// butterfly order, logit = a1 + a2, expf(x - max) / sum), so for h = 3 or 4 (the softmax operator then sums in the
// same order) the result is bit-identical to the operator chain, otherwise equal up to the order of that sum.
// A query's logits stay in registers (lane (p, c): head c of pair slot p, one register per wave pass) when it
// has at most 16 * WL_MAXP pairs; longer rows park them in the output buffer instead.
// ------------------------------------------------------------------------------------------------
constexpr int WL_MAXP = 8;

template <int HG>
__global__ __launch_bounds__(768, 6) void wlogit_softmax_kernel(int N, int h, int L, const float *__restrict__ q,
                                                                const int *__restrict__ offs, const float *__restrict__ k,
                                                                const int *__restrict__ idx_k, const float *__restrict__ table_q,
                                                                const float *__restrict__ table_k, const int *__restrict__ rel,
                                                                float *__restrict__ attn, const int *__restrict__ rord) {
    constexpr int D = 16;
    P2_WALK_PROLOGUE
    float *Tq = lds, *Tk = lds + tsz;
    stage_table<D>(Tq, table_q, L, h, h0, hgn);
    stage_table<D>(Tk, table_k, L, h, h0, hgn);
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    // max / sum over the pair slots of a head: lanes that differ in the bits above LPG
    auto slots_max = [&](float v) {
        for (int st = LPG; st < 64; st <<= 1) v = fmaxf(v, __shfl_xor(v, st, 64));
        return v;
    };
    auto slots_sum = [&](float v) {
        for (int st = LPG; st < 64; st <<= 1) v += __shfl_xor(v, st, 64);
        return v;
    };
    for (RowSlots slots(rord, N, wpb, wave); slots.more(); slots.next()) {
        const int qi = slots.row();
        float4 q4[HG];
#pragma unroll
        for (int t = 0; t < HG; t++)
            q4[t] = ldg4(q + (size_t)qi * C + (h0 + min(t, hgn - 1)) * D + 4 * c);
        const int s = offs[qi], e = offs[qi + 1];
        if (e <= s) continue;
        const int np = (e - s + PPW - 1) / PPW;
        // logit of head c (lane c of the pair's lane group keeps it) for pair m
        auto logit = [&](int m, bool valid) -> float {
            const int mm = valid ? m : s;
            const int j = idx_k[mm];
            const int r0 = clampr(rel[mm * 3 + 0], L), r1 = clampr(rel[mm * 3 + 1], L), r2 = clampr(rel[mm * 3 + 2], L);
            float keep = 0.f;
            float4 k4[HG];
#pragma unroll
            for (int t = 0; t < HG; t++) k4[t] = ldg4(k + (size_t)j * C + (h0 + min(t, hgn - 1)) * D + 4 * c);  // no per-head guards
#pragma unroll
            for (int t = 0; t < HG; t++) {
                const int te = min(t, hgn - 1);
                const float a1 = xor_sum<1, LPG>(dot4(q4[t], k4[t]));
                const float a2 = xor_sum<1, LPG>(dot4(q4[t], tsum<D>(Tq, L, te, r0, r1, r2, c)) + dot4(k4[t], tsum<D>(Tk, L, te, r0, r1, r2, c)));
                if (c == t) keep = a1 + a2;
            }
            return keep;
        };
        const bool mine = c < hgn;
        if (np <= WL_MAXP) {
            float lg[WL_MAXP];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < WL_MAXP; i++) {
                lg[i] = -INFINITY;
                if (i < np) {  // wave-uniform
                    const int m = s + i * PPW + p;
                    const float v = logit(m, m < e);
                    if (m < e && mine) lg[i] = v;
                    mx = fmaxf(mx, lg[i]);
                }
            }
            mx = slots_max(mx);
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < WL_MAXP; i++) {
                if (i < np) {
                    const int m = s + i * PPW + p;
                    if (m < e && mine) {
                        lg[i] = expf(lg[i] - mx);
                        sum += lg[i];
                    }
                }
            }
            sum = slots_sum(sum);
#pragma unroll
            for (int i = 0; i < WL_MAXP; i++) {
                if (i < np) {
                    const int m = s + i * PPW + p;
                    if (m < e && mine) attn[(size_t)m * h + h0 + c] = lg[i] / sum;
                }
            }
        } else {
            float mx = -INFINITY;
            for (int m0 = s; m0 < e; m0 += PPW) {
                const int m = m0 + p;
                const float v = logit(m, m < e);
                if (m < e && mine) {
                    attn[(size_t)m * h + h0 + c] = v;
                    mx = fmaxf(mx, v);
                }
            }
            mx = slots_max(mx);
            float sum = 0.f;
            for (int m = s + p; m < e; m += PPW)
                if (mine) {
                    const float ex = expf(attn[(size_t)m * h + h0 + c] - mx);  // written by this very lane above
                    attn[(size_t)m * h + h0 + c] = ex;
                    sum += ex;
                }
            sum = slots_sum(sum);
            for (int m = s + p; m < e; m += PPW)
                if (mine) attn[(size_t)m * h + h0 + c] = attn[(size_t)m * h + h0 + c] / sum;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A2 backward, by query: grad_q (stored), grad_table_q (LDS -> atomics).
// KEYSIDE: also grad_k (global atomics) and grad_table_k — the no-CSC fallback.
// ------------------------------------------------------------------------------------------------
template <int D, int HG, bool KEYSIDE>
__global__ __launch_bounds__(256) void a2_bwd_query_kernel(int N, int h, int L, const float *__restrict__ go,
                                                           const float *__restrict__ q, const int *__restrict__ offs,
                                                           const float *__restrict__ k, const int *__restrict__ idx_k,
                                                           const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                           const int *__restrict__ rel, float *__restrict__ grad_q,
                                                           float *__restrict__ grad_k, float *__restrict__ grad_table_q,
                                                           float *__restrict__ grad_table_k) {
    P2_WALK_PROLOGUE
    float *Tq = lds, *Gq = lds + tsz, *Tk = lds + 2 * tsz, *Gk = lds + 3 * tsz;
    stage_table<D>(Tq, table_q, L, h, h0, hgn);
    zero_lds<D>(Gq, tsz);
    if (KEYSIDE) {
        stage_table<D>(Tk, table_k, L, h, h0, hgn);
        zero_lds<D>(Gk, tsz);
    }
    __syncthreads();
    for (int qi = blockIdx.x * 4 + wave; qi < N; qi += gridDim.x * 4) {
        float4 q4[HG], acc[HG];
#pragma unroll
        for (int t = 0; t < HG; t++) {
            q4[t] = t < hgn ? ldg4(q + (size_t)qi * C + (h0 + t) * D + 4 * c) : make_float4(0, 0, 0, 0);
            acc[t] = make_float4(0, 0, 0, 0);
        }
        const int s = offs[qi], e = offs[qi + 1];
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int m = m0 + p;
            if (m < e) {
                const int r0 = clampr(rel[m * 3 + 0], L), r1 = clampr(rel[m * 3 + 1], L), r2 = clampr(rel[m * 3 + 2], L);
                const int j = KEYSIDE ? idx_k[m] : 0;
#pragma unroll
                for (int t = 0; t < HG; t++) {
                    if (t < hgn) {
                        const float g = go[(size_t)m * h + h0 + t];
                        acc[t] = fma4(g, tsum<D>(Tq, L, t, r0, r1, r2, c), acc[t]);
                        tadd<D>(Gq, L, t, r0, r1, r2, c, scale4(g, q4[t]));
                        if (KEYSIDE) {
                            float *kp = (float *)k + (size_t)j * C + (h0 + t) * D + 4 * c;
                            const float4 k4 = ldg4(kp);
                            const float4 gk = scale4(g, tsum<D>(Tk, L, t, r0, r1, r2, c));
                            float *d = grad_k + (size_t)j * C + (h0 + t) * D + 4 * c;
                            atomicAdd(d + 0, gk.x); atomicAdd(d + 1, gk.y); atomicAdd(d + 2, gk.z); atomicAdd(d + 3, gk.w);
                            tadd<D>(Gk, L, t, r0, r1, r2, c, scale4(g, k4));
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < HG; t++) {
            if (t < hgn) {
                float4 tot = xor_sum4<LPG, 64>(acc[t]);
                if (p == 0) stg4(grad_q + (size_t)qi * C + (h0 + t) * D + 4 * c, tot);
            }
        }
    }
    __syncthreads();
    flush_table<D>(Gq, grad_table_q, L, h, h0, hgn);
    if (KEYSIDE) flush_table<D>(Gk, grad_table_k, L, h, h0, hgn);
}

// A2 backward, by key (CSC): grad_k (accumulated into the pre-zeroed buffer), grad_table_k.
template <int D, int HG>
__global__ __launch_bounds__(256) void a2_bwd_key_kernel(int N, int h, int L, const float *__restrict__ go,
                                                         const float *__restrict__ k, const int *__restrict__ csc_offs,
                                                         const int *__restrict__ csc_pair, const float *__restrict__ table_k,
                                                         const int *__restrict__ rel, float *__restrict__ grad_k,
                                                         float *__restrict__ grad_table_k) {
    P2_WALK_PROLOGUE
    float *Tk = lds, *Gk = lds + tsz;
    stage_table<D>(Tk, table_k, L, h, h0, hgn);
    zero_lds<D>(Gk, tsz);
    __syncthreads();
    for (int kj = blockIdx.x * 4 + wave; kj < N; kj += gridDim.x * 4) {
        float4 k4[HG], acc[HG];
#pragma unroll
        for (int t = 0; t < HG; t++) {
            k4[t] = t < hgn ? ldg4(k + (size_t)kj * C + (h0 + t) * D + 4 * c) : make_float4(0, 0, 0, 0);
            acc[t] = make_float4(0, 0, 0, 0);
        }
        const int s = csc_offs[kj], e = csc_offs[kj + 1];
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int slot = m0 + p;
            if (slot < e) {
                const int m = csc_pair[slot];
                const int r0 = clampr(rel[m * 3 + 0], L), r1 = clampr(rel[m * 3 + 1], L), r2 = clampr(rel[m * 3 + 2], L);
#pragma unroll
                for (int t = 0; t < HG; t++) {
                    if (t < hgn) {
                        const float g = go[(size_t)m * h + h0 + t];
                        acc[t] = fma4(g, tsum<D>(Tk, L, t, r0, r1, r2, c), acc[t]);
                        tadd<D>(Gk, L, t, r0, r1, r2, c, scale4(g, k4[t]));
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < HG; t++) {
            if (t < hgn) {
                float4 tot = xor_sum4<LPG, 64>(acc[t]);
                if (p == 0) {
                    float *o = grad_k + (size_t)kj * C + (h0 + t) * D + 4 * c;
                    stg4(o, add4(tot, ldg4(o)));
                }
            }
        }
    }
    __syncthreads();
    flush_table<D>(Gk, grad_table_k, L, h, h0, hgn);
}

// ------------------------------------------------------------------------------------------------
// A4 forward: out[q,hh,:] = sum_m attn[m,hh] * (v[idx1[m],hh,:] + Tv(m,hh,:))
// ------------------------------------------------------------------------------------------------
template <int D, int HG>
__global__ __launch_bounds__(512, 6) void a4_fwd_kernel(int N, int h, int L, const float *__restrict__ attn,
                                                        const float *__restrict__ v, const int *__restrict__ offs,
                                                        const int *__restrict__ idx1, const float *__restrict__ table,
                                                        const int *__restrict__ rel, float *__restrict__ out, const int *__restrict__ rord) {
    P2_WALK_PROLOGUE
    float *Tv = lds;
    stage_table<D>(Tv, table, L, h, h0, hgn);
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    for (RowSlots slots(rord, N, wpb, wave); slots.more(); slots.next()) {
        const int qi = slots.row();
        float4 acc[HG];
#pragma unroll
        for (int t = 0; t < HG; t++) acc[t] = make_float4(0, 0, 0, 0);
        const int s = offs[qi], e = offs[qi + 1];
        // no per-head guards (a2_fwd_kernel): a slot past the group's last head repeats it, its sum is not stored.
        // The ids of the next pass are requested together with the weights and value rows of this one (one round
        // trip per pass); a slot past the row's end repeats the row's last pair with weight zero.
        PairIds nx = load_pair_ids(idx1, rel, max(0, min(s + p, e - 1)));
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int m = m0 + p;
            const int mm = min(m, e - 1);
            const PairIds cur = nx;
            nx = load_pair_ids(idx1, rel, min(m + PPW, e - 1));
            float a[HG];
            float4 v4[HG];
#pragma unroll
            for (int t = 0; t < HG; t++) {
                a[t] = attn[(size_t)mm * h + h0 + min(t, hgn - 1)];
                v4[t] = ldg4(v + (size_t)cur.j * C + (h0 + min(t, hgn - 1)) * D + 4 * c);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int r0 = clampr(cur.q0, L), r1 = clampr(cur.q1, L), r2 = clampr(cur.q2, L);
            if (m < e) {
#pragma unroll
                for (int t = 0; t < HG; t++) acc[t] = fma4(a[t], add4(tsum<D>(Tv, L, min(t, hgn - 1), r0, r1, r2, c), v4[t]), acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < HG; t++) {
            if (t < hgn) {
                float4 tot = xor_sum4<LPG, 64>(acc[t]);
                if (p == 0) stg4(out + (size_t)qi * C + (h0 + t) * D + 4 * c, tot);
            }
        }
    }
}

// A4 backward, by query: grad_attn (stored), grad_table (LDS -> atomics); KEYSIDE: grad_v by atomics.
template <int D, int HG, bool KEYSIDE>
__global__ __launch_bounds__(256) void a4_bwd_query_kernel(int N, int h, int L, const float *__restrict__ go,
                                                           const int *__restrict__ offs, const int *__restrict__ idx1,
                                                           const float *__restrict__ attn, const float *__restrict__ v,
                                                           const float *__restrict__ table, const int *__restrict__ rel,
                                                           float *__restrict__ grad_attn, float *__restrict__ grad_v,
                                                           float *__restrict__ grad_table) {
    P2_WALK_PROLOGUE
    float *Tv = lds, *Gv = lds + tsz;
    stage_table<D>(Tv, table, L, h, h0, hgn);
    zero_lds<D>(Gv, tsz);
    __syncthreads();
    for (int qi = blockIdx.x * 4 + wave; qi < N; qi += gridDim.x * 4) {
        float4 g4[HG];
#pragma unroll
        for (int t = 0; t < HG; t++)
            g4[t] = t < hgn ? ldg4(go + (size_t)qi * C + (h0 + t) * D + 4 * c) : make_float4(0, 0, 0, 0);
        const int s = offs[qi], e = offs[qi + 1];
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int m = m0 + p;
            const bool valid = m < e;
            const int mm = valid ? m : s;
            const int j = idx1[mm];
            const int r0 = clampr(rel[mm * 3 + 0], L), r1 = clampr(rel[mm * 3 + 1], L), r2 = clampr(rel[mm * 3 + 2], L);
            float keep = 0.f;
#pragma unroll
            for (int t = 0; t < HG; t++) {
                if (t < hgn) {
                    const float4 v4 = ldg4(v + (size_t)j * C + (h0 + t) * D + 4 * c);
                    float part = dot4(add4(tsum<D>(Tv, L, t, r0, r1, r2, c), v4), g4[t]);
                    float tot = xor_sum<1, LPG>(part);
                    if (c == t) keep = tot;
                    if (valid) {
                        const float a = attn[(size_t)m * h + h0 + t];
                        const float4 ag = scale4(a, g4[t]);
                        tadd<D>(Gv, L, t, r0, r1, r2, c, ag);
                        if (KEYSIDE) {
                            float *d = grad_v + (size_t)j * C + (h0 + t) * D + 4 * c;
                            atomicAdd(d + 0, ag.x); atomicAdd(d + 1, ag.y); atomicAdd(d + 2, ag.z); atomicAdd(d + 3, ag.w);
                        }
                    }
                }
            }
            if (valid && c < hgn) grad_attn[(size_t)m * h + h0 + c] = keep;
        }
    }
    __syncthreads();
    flush_table<D>(Gv, grad_table, L, h, h0, hgn);
}

// key-major accumulate used for grad_v (defined in attention.hip, instantiated here through a thin
// duplicate to keep the translation units independent)
template <int D>
__global__ __launch_bounds__(256) void key_accum_kernel(int N, int h, const int *__restrict__ offs,
                                                        const int *__restrict__ sidx, const int *__restrict__ widx,
                                                        const float *__restrict__ w, const float *__restrict__ src,
                                                        float *__restrict__ out, const int *__restrict__ rord) {
    constexpr int LPG = Geo<D>::LPG, PPW = Geo<D>::PPW, HC = 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const RowSlots slots(rord, N, 4, wave);
    if (!slots.more()) return;
    const int row = slots.row();
    const int C = h * D;
    const int p = lane / LPG, c = lane % LPG;
    const int s = offs[row], e = offs[row + 1];
    for (int hb = blockIdx.y * HC; hb < h; hb += gridDim.y * HC) {  // head chunks over blockIdx.y (attention.hip)
        float4 acc[HC];
#pragma unroll
        for (int t = 0; t < HC; t++) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 prev[HC];  // the row's running sum, requested now and consumed after the walk
#pragma unroll
        for (int t = 0; t < HC; t++) prev[t] = ldg4(out + (size_t)row * C + min(hb + t, h - 1) * D + 4 * c);
        // the ids of the next pass are requested together with the weights and rows of this one (one round trip per
        // pass); a slot past the row's end repeats the row's last slot and is not added
        const int first = max(0, min(s + p, e - 1));
        int sn = sidx[first], wn = widx[first];
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int slot = m0 + p;
            const int nslot = min(slot + PPW, e - 1);
            const float *srow = src + (size_t)sn * C + 4 * c;
            const float *wrow = w + (size_t)wn * h;
            sn = sidx[nslot];
            wn = widx[nslot];
            // no per-head guards: a slot past the last head repeats it (its sum is not stored), so the HC weight and
            // row loads of a pass are issued together instead of one dependent round trip per head
            if (slot < e) {
                float wv[HC];
                float4 sv[HC];
#pragma unroll
                for (int t = 0; t < HC; t++) {
                    const int hh = min(hb + t, h - 1);
                    wv[t] = wrow[hh];
                    sv[t] = ldg4(srow + hh * D);
                }
                __builtin_amdgcn_sched_barrier(0);  // every load of the pass before its first use
#pragma unroll
                for (int t = 0; t < HC; t++) acc[t] = fma4(wv[t], sv[t], acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < HC; t++) {
            if (hb + t < h) {
                float4 tot = xor_sum4<LPG, 64>(acc[t]);
                if (p == 0) stg4(out + (size_t)row * C + (hb + t) * D + 4 * c, add4(tot, prev[t]));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v1 pair-indexed forms (relative_pos_encoding_cuda_kernel.cu:7-136): one thread per (pair, head),
// atomics where the reference has them.  Tables are read from global memory (L is not needed).
// ------------------------------------------------------------------------------------------------
__global__ void dot_v1_fwd_kernel(int M, int h, int d, const float *q, const int *index, const float *table,
                                  const int *rel, float *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h;
    const float *qv = q + ((size_t)index[m] * h + hh) * d;
    float sum = 0.f;
    for (int ax = 0; ax < 3; ax++) {
        const float *tb = table + (((size_t)rel[m * 3 + ax] * h + hh) * d) * 3 + ax;
        for (int i = 0; i < d; i++) sum = fmaf(qv[i], tb[i * 3], sum);
    }
    out[t] += sum;
}
__global__ void dot_v1_bwd_kernel(int M, int h, int d, const float *go, const float *q, const int *index,
                                  const float *table, const int *rel, float *gq, float *gt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h;
    const size_t qb = ((size_t)index[m] * h + hh) * d;
    const float g = go[t];
    for (int ax = 0; ax < 3; ax++) {
        const size_t tb = (((size_t)rel[m * 3 + ax] * h + hh) * d) * 3 + ax;
        for (int i = 0; i < d; i++) {
            atomicAdd(gq + qb + i, g * table[tb + i * 3]);
            atomicAdd(gt + tb + i * 3, g * q[qb + i]);
        }
    }
}
__global__ void av_v1_fwd_kernel(int M, int h, int d, const float *attn, const float *v, const int *i0, const int *i1,
                                 const float *table, const int *rel, float *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h;
    const size_t ob = ((size_t)i0[m] * h + hh) * d, vb = ((size_t)i1[m] * h + hh) * d;
    const float a = attn[t];
    const float *t0 = table + (((size_t)rel[m * 3 + 0] * h + hh) * d) * 3 + 0;
    const float *t1 = table + (((size_t)rel[m * 3 + 1] * h + hh) * d) * 3 + 1;
    const float *t2 = table + (((size_t)rel[m * 3 + 2] * h + hh) * d) * 3 + 2;
    for (int i = 0; i < d; i++) atomicAdd(out + ob + i, a * (v[vb + i] + (t0[i * 3] + t1[i * 3] + t2[i * 3])));
}
__global__ void av_v1_bwd_kernel(int M, int h, int d, const float *go, const int *i0, const int *i1, const float *attn,
                                 const float *v, const float *table, const int *rel, float *ga, float *gv, float *gt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * h) return;
    const int m = t / h, hh = t % h;
    const size_t ob = ((size_t)i0[m] * h + hh) * d, vb = ((size_t)i1[m] * h + hh) * d;
    const float a = attn[t];
    const size_t b0 = (((size_t)rel[m * 3 + 0] * h + hh) * d) * 3 + 0;
    const size_t b1 = (((size_t)rel[m * 3 + 1] * h + hh) * d) * 3 + 1;
    const size_t b2 = (((size_t)rel[m * 3 + 2] * h + hh) * d) * 3 + 2;
    float sum = 0.f;
    for (int i = 0; i < d; i++) {
        const float g = go[ob + i];
        sum = fmaf(g, v[vb + i] + (table[b0 + i * 3] + table[b1 + i * 3] + table[b2 + i * 3]), sum);
        atomicAdd(gv + vb + i, g * a);
        atomicAdd(gt + b0 + i * 3, g * a);
        atomicAdd(gt + b1 + i * 3, g * a);
        atomicAdd(gt + b2 + i * 3, g * a);
    }
    ga[t] += sum;
}

// chooses heads-per-workgroup so that `narr` table images fit the LDS budget
static int pick_hg(int h, int L, int D, int narr) {
    const size_t per_head = (size_t)narr * 3 * L * D * sizeof(float);
    int hg = (int)(kLdsBudget / per_head);
    if (hg < 1) {
        if (per_head <= 150 * 1024) return 1;  // one workgroup per CU
        return 0;
    }
    if (hg > 3) hg = 3;
    if (hg > h) hg = h;
    return hg;
}

static int table_rows_or_error() {
    const int L = state().table_rows;
    if (L <= 0) set_error("rel-pos tables: call pointops2_set_table_rows(L) before this launcher");
    return L;
}

// waves_per_block: rows a workgroup handles at a time; per_cu: resident workgroups per CU (LDS / register budget)
static int persistent_blocks(int rows, int head_groups, int waves_per_block = 4, int per_cu = 2) {
    int want = div_up(rows, waves_per_block);
    const int cus = usable_cus();
    int cap = cus * per_cu;
    if (head_groups > 1) cap = max(cus * per_cu / head_groups, cus / 2);
    return min(want, cap);
}

template <int D, typename F>
static void with_hg(int hg, F f) {
    if (hg == 1) f(std::integral_constant<int, 1>{});
    else if (hg == 2) f(std::integral_constant<int, 2>{});
    else f(std::integral_constant<int, 3>{});
}

}  // namespace p2

using namespace p2;

#define P2_LAUNCH_HG(D_, narr_, BODY)                                                           \
    {                                                                                           \
        const int hg = pick_hg(h, L, D_, narr_);                                                \
        if (hg == 0) { set_error("rel-pos table slice does not fit in LDS"); return; }          \
        const int ngroups = div_up(h, hg);                                                      \
        const size_t lds_bytes = (size_t)narr_ * hg * 3 * L * D_ * sizeof(float);               \
        with_hg<D_>(hg, [&](auto hgtag) {                                                       \
            constexpr int HGc = decltype(hgtag)::value;                                         \
            constexpr int Dc = D_;                                                              \
            (void)HGc; (void)Dc;                                                                \
            BODY                                                                                \
        });                                                                                     \
        (void)ngroups; (void)lds_bytes;                                                         \
    }

extern "C" {

void dot_prod_with_idx_forward_cuda_launcher_v3(int N, int M, int h, int hdim, int n_max, const float *q,
                                                const int *index_q_offsets, const float *k, const int *index_k,
                                                const float *table_q, const float *table_k, const int *rel_idx,
                                                float *output) {
    (void)n_max;
    if (N <= 0 || M <= 0) return;
    const int L = state().table_rows;
    if (L <= 0) {  // table length not announced: the generic global-memory kernels (rpe_fallback.hip)
        a2_fwd_global(N, M, h, hdim, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output);
        check_launch();
        return;
    }
    hipStream_t st = state().stream;
    if (hdim == 16) P2_LAUNCH_HG(16, 2, {
        allow_big_lds(a2_fwd_kernel<Dc, HGc>, lds_bytes);
        hipLaunchKernelGGL((a2_fwd_kernel<Dc, HGc>), dim3(persistent_blocks(N, ngroups, 12, 2), ngroups), dim3(768), lds_bytes, st,
                           N, h, L, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output, rows_in_order(N));
    })
    else if (hdim == 32) P2_LAUNCH_HG(32, 2, {
        allow_big_lds(a2_fwd_kernel<Dc, HGc>, lds_bytes);
        hipLaunchKernelGGL((a2_fwd_kernel<Dc, HGc>), dim3(persistent_blocks(N, ngroups, 12, 2), ngroups), dim3(768), lds_bytes, st,
                           N, h, L, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output, rows_in_order(N));
    })
    else { set_error("d != 16 and d != 32"); return; }
    check_launch();
}

void window_logits_softmax_forward_launcher(int N, int M, int h, int hdim, const float *q, const int *index_q_offsets,
                                             const float *k, const int *index_k, const float *table_q, const float *table_k,
                                             const int *rel_idx, float *attn) {
    if (N <= 0 || M <= 0) return;
    const int L = table_rows_or_error();
    if (L <= 0) return;
    if (hdim != 16) { set_error("window_logits_softmax: d != 16"); return; }
    hipStream_t st = state().stream;
    P2_LAUNCH_HG(16, 2, {
        allow_big_lds(wlogit_softmax_kernel<HGc>, lds_bytes);
        hipLaunchKernelGGL((wlogit_softmax_kernel<HGc>), dim3(persistent_blocks(N, ngroups, 12, 2), ngroups), dim3(768), lds_bytes, st,
                           N, h, L, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, attn, rows_in_order(N));
    })
    check_launch();
}

void window_attention_backward_launcher(int N, int M, int h, int hdim, const float *grad_out, const float *q, const float *k,
                                        const float *v, const float *attn, const int *index0_offsets, const int *index1,
                                        const float *table_q, const float *table_k, const float *table_v, const int *rel_idx,
                                        float *grad_logit, float *grad_q, float *grad_k, float *grad_v, float *grad_table_q,
                                        float *grad_table_k, float *grad_table_v) {
    if (N <= 0 || M <= 0) return;
    const int L = table_rows_or_error();
    if (L <= 0) return;
    const LaunchState &ls = state();
    const int NK = ls.key_rows > 0 ? ls.key_rows : N;
    if (!wattn_bwd(N, NK, M, h, hdim, L, grad_out, q, k, v, attn, index0_offsets, index1, table_q, table_k, table_v, rel_idx,
                   ls.csc_offsets, ls.csc_pair, ls.csc_query, grad_logit, grad_q, grad_k, grad_v, grad_table_q, grad_table_k,
                   grad_table_v)) {
        set_error("window_attention_backward: needs d = 16, L <= 80 and a key-major view (pointops2_set_csc)");
        return;
    }
    check_launch();
}

void dot_prod_with_idx_backward_cuda_launcher_v3(int N, int M, int h, int hdim, int n_max, const float *grad_out,
                                                 const float *q, const int *index_q_offsets, const float *k,
                                                 const int *index_k, const float *table_q, const float *table_k,
                                                 const int *rel_idx, float *grad_q, float *grad_k,
                                                 float *grad_table_q, float *grad_table_k) {
    (void)n_max;
    if (N <= 0 || M <= 0) return;
    const int L = state().table_rows;
    if (L <= 0) {
        a2_bwd_global(N, M, h, hdim, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, grad_q, grad_k, grad_table_q, grad_table_k);
        check_launch();
        return;
    }
    hipStream_t st = state().stream;
    const LaunchState &ls = state();
    const int *co = ls.csc_offsets, *cp = ls.csc_pair;
    const int NK = ls.key_rows > 0 ? ls.key_rows : N;
    // D=16, L<=80 with a CSC view: table gradients on the matrix cores (rpe_bwd_mfma.hip)
    if (a2_bwd_mfma(N, NK, M, h, hdim, L, grad_out, q, index_q_offsets, k, table_q, table_k, rel_idx, co, cp, grad_q, grad_k,
                    grad_table_q, grad_table_k)) {
        check_launch();
        return;
    }
#define P2_A2_BWD(D_)                                                                                                       \
    if (co) {                                                                                                               \
        P2_LAUNCH_HG(D_, 2, {                                                                                               \
            allow_big_lds(a2_bwd_query_kernel<Dc, HGc, false>, lds_bytes);                                                  \
            allow_big_lds(a2_bwd_key_kernel<Dc, HGc>, lds_bytes);                                                           \
            hipLaunchKernelGGL((a2_bwd_query_kernel<Dc, HGc, false>), dim3(persistent_blocks(N, ngroups), ngroups), dim3(256), \
                               lds_bytes, st, N, h, L, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx,  \
                               grad_q, grad_k, grad_table_q, grad_table_k);                                                 \
            hipLaunchKernelGGL((a2_bwd_key_kernel<Dc, HGc>), dim3(persistent_blocks(NK, ngroups), ngroups), dim3(256),       \
                               lds_bytes, st, NK, h, L, grad_out, k, co, cp, table_k, rel_idx, grad_k, grad_table_k);       \
        })                                                                                                                  \
    } else {                                                                                                                \
        P2_LAUNCH_HG(D_, 4, {                                                                                               \
            allow_big_lds(a2_bwd_query_kernel<Dc, HGc, true>, lds_bytes);                                                   \
            hipLaunchKernelGGL((a2_bwd_query_kernel<Dc, HGc, true>), dim3(persistent_blocks(N, ngroups), ngroups), dim3(256), \
                               lds_bytes, st, N, h, L, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx,  \
                               grad_q, grad_k, grad_table_q, grad_table_k);                                                 \
        })                                                                                                                  \
    }
    if (hdim == 16) { P2_A2_BWD(16) }
    else if (hdim == 32) { P2_A2_BWD(32) }
    else { set_error("d != 16 and d != 32"); return; }
#undef P2_A2_BWD
    check_launch();
}

void attention_step2_with_rel_pos_value_forward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max,
                                                                 const float *attn, const float *v,
                                                                 const int *index0_offsets, const int *index1,
                                                                 const float *table, const int *rel_idx, float *output) {
    (void)n_max;
    if (N <= 0 || M <= 0) return;
    const int L = state().table_rows;
    if (L <= 0) {
        a4_fwd_global(N, M, h, hdim, attn, v, index0_offsets, index1, table, rel_idx, output);
        check_launch();
        return;
    }
    hipStream_t st = state().stream;
    if (hdim == 16) P2_LAUNCH_HG(16, 1, {
        allow_big_lds(a4_fwd_kernel<Dc, HGc>, lds_bytes);
        hipLaunchKernelGGL((a4_fwd_kernel<Dc, HGc>), dim3(persistent_blocks(N, ngroups, 8, 3), ngroups), dim3(512), lds_bytes, st,
                           N, h, L, attn, v, index0_offsets, index1, table, rel_idx, output, rows_in_order(N));
    })
    else if (hdim == 32) P2_LAUNCH_HG(32, 1, {
        allow_big_lds(a4_fwd_kernel<Dc, HGc>, lds_bytes);
        hipLaunchKernelGGL((a4_fwd_kernel<Dc, HGc>), dim3(persistent_blocks(N, ngroups, 8, 3), ngroups), dim3(512), lds_bytes, st,
                           N, h, L, attn, v, index0_offsets, index1, table, rel_idx, output, rows_in_order(N));
    })
    else { set_error("d != 16 and d != 32"); return; }
    check_launch();
}

void attention_step2_with_rel_pos_value_backward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max,
                                                                  const float *grad_out, const int *index0_offsets,
                                                                  const int *index1, const float *attn, const float *v,
                                                                  const float *table, const int *rel_idx,
                                                                  float *grad_attn, float *grad_v, float *grad_table) {
    (void)n_max;
    if (N <= 0 || M <= 0) return;
    const int L = state().table_rows;
    if (L <= 0) {
        a4_bwd_global(N, M, h, hdim, grad_out, index0_offsets, index1, attn, v, table, rel_idx, grad_attn, grad_v, grad_table);
        check_launch();
        return;
    }
    hipStream_t st = state().stream;
    const LaunchState &ls = state();
    const int *co = ls.csc_offsets, *cp = ls.csc_pair, *cq = ls.csc_query;
    const int NK4 = ls.key_rows > 0 ? ls.key_rows : N;
    if (co && hdim == 16 && L <= 80) {
        ForkJoin fj(st, fork_worthwhile((int64_t)M * h));  // grad_attn, grad_v and grad_table are independent
        const int *kord = rows_in_order(NK4);
        hipLaunchKernelGGL(key_accum_kernel<16>, dim3(kord ? ordered_grid(NK4, 4) : div_up(NK4, 4), div_up(h, 4)), dim3(256), 0, fj.lane(2), NK4, h, co, cq, cp, attn, grad_out, grad_v, kord);
        a4_bwd_mfma(N, h, hdim, L, grad_out, index0_offsets, index1, attn, v, table, rel_idx, grad_attn, grad_table, fj);
        check_launch();
        return;
    }
#define P2_A4_BWD(D_)                                                                                                       \
    P2_LAUNCH_HG(D_, 2, {                                                                                                   \
        allow_big_lds(a4_bwd_query_kernel<Dc, HGc, false>, lds_bytes);                                                      \
        allow_big_lds(a4_bwd_query_kernel<Dc, HGc, true>, lds_bytes);                                                       \
        if (co) {                                                                                                           \
            hipLaunchKernelGGL((a4_bwd_query_kernel<Dc, HGc, false>), dim3(persistent_blocks(N, ngroups), ngroups), dim3(256), \
                               lds_bytes, st, N, h, L, grad_out, index0_offsets, index1, attn, v, table, rel_idx, grad_attn, \
                               grad_v, grad_table);                                                                         \
            hipLaunchKernelGGL(key_accum_kernel<Dc>, dim3(div_up(NK4, 4), div_up(h, 4)), dim3(256), 0, st, NK4, h, co, cq, cp, attn, grad_out, \
                               grad_v, (const int *)nullptr);                                                                                     \
        } else {                                                                                                            \
            hipLaunchKernelGGL((a4_bwd_query_kernel<Dc, HGc, true>), dim3(persistent_blocks(N, ngroups), ngroups), dim3(256), \
                               lds_bytes, st, N, h, L, grad_out, index0_offsets, index1, attn, v, table, rel_idx, grad_attn, \
                               grad_v, grad_table);                                                                         \
        }                                                                                                                   \
    })
    if (hdim == 16) { P2_A4_BWD(16) }
    else if (hdim == 32) { P2_A4_BWD(32) }
    else { set_error("d != 16 and d != 32"); return; }
#undef P2_A4_BWD
    check_launch();
}

// ---- v1 forms ----
void dot_prod_with_idx_forward_cuda_launcher(int N, int M, int h, int hdim, const float *q, const int *index,
                                             const float *table, const int *rel_idx, float *output) {
    (void)N;
    if (M <= 0) return;
    hipLaunchKernelGGL(dot_v1_fwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, hdim, q, index, table, rel_idx, output);
    check_launch();
}
void dot_prod_with_idx_backward_cuda_launcher(int N, int M, int h, int hdim, const float *grad_out,
                                              const float *q, const int *index, const float *table,
                                              const int *rel_idx, float *grad_q, float *grad_table) {
    (void)N;
    if (M <= 0) return;
    hipLaunchKernelGGL(dot_v1_bwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, hdim, grad_out, q, index, table, rel_idx, grad_q, grad_table);
    check_launch();
}
void attention_step2_with_rel_pos_value_forward_cuda_launcher(int N, int M, int h, int hdim, const float *attn,
                                                              const float *v, const int *index0, const int *index1,
                                                              const float *table, const int *rel_idx, float *output) {
    (void)N;
    if (M <= 0) return;
    hipLaunchKernelGGL(av_v1_fwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, hdim, attn, v, index0, index1, table, rel_idx, output);
    check_launch();
}
void attention_step2_with_rel_pos_value_backward_cuda_launcher(int N, int M, int h, int hdim, const float *grad_out,
                                                               const int *index0, const int *index1, const float *attn,
                                                               const float *v, const float *table, const int *rel_idx,
                                                               float *grad_attn, float *grad_v, float *grad_table) {
    (void)N;
    if (M <= 0) return;
    hipLaunchKernelGGL(av_v1_bwd_kernel, dim3(div_up64((int64_t)M * h, 256)), dim3(256), 0, state().stream, M, h, hdim, grad_out, index0, index1, attn, v, table, rel_idx, grad_attn, grad_v, grad_table);
    check_launch();
}

// The bucketed "v2" bias (relative_pos_encoding_cuda_kernel_v2.cu:9-243) computes the same function
// as v3; its extra arguments (T, rel_idx_offsets, sort_indices: pairs bucketed by merged rel index)
// only drive the reference's work distribution.  index_q/index_k are pair-indexed (unsorted), so it
// is served by two passes of the v1 single-table kernel (q-side + k-side), which is exactly the
// identity the reference's own test checks (test_relative_pos_encoding_op_step1_v3.py:60-62).
void dot_prod_with_idx_forward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max, int T, const float *q,
                                                const int *index_q, const float *k, const int *index_k,
                                                const float *table_q, const float *table_k, const int *rel_idx,
                                                const int *rel_idx_offsets, const int *sort_indices, float *output) {
    (void)n_max; (void)T; (void)rel_idx_offsets; (void)sort_indices;
    dot_prod_with_idx_forward_cuda_launcher(N, M, h, hdim, q, index_q, table_q, rel_idx, output);
    dot_prod_with_idx_forward_cuda_launcher(N, M, h, hdim, k, index_k, table_k, rel_idx, output);
}
void dot_prod_with_idx_backward_cuda_launcher_v2(int N, int M, int h, int hdim, int n_max, int T, const float *grad_out,
                                                 const float *q, const int *index_q, const float *k, const int *index_k,
                                                 const float *table_q, const float *table_k, const int *rel_idx,
                                                 const int *rel_idx_offsets, const int *sort_indices, float *grad_q,
                                                 float *grad_k, float *grad_table_q, float *grad_table_k) {
    (void)n_max; (void)T; (void)rel_idx_offsets; (void)sort_indices;
    dot_prod_with_idx_backward_cuda_launcher(N, M, h, hdim, grad_out, q, index_q, table_q, rel_idx, grad_q, grad_table_q);
    dot_prod_with_idx_backward_cuda_launcher(N, M, h, hdim, grad_out, k, index_k, table_k, rel_idx, grad_k, grad_table_k);
}

}  // extern "C"
