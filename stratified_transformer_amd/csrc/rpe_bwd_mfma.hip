// Backward of A2 / A4 with the table gradients on the matrix cores, gfx950.
//
// The gradient of a relative-position table is
//     grad_table[r, hh, i, ax] = sum over pairs m with rel[m, ax] == r of  w[m, hh] * X[row(m), hh, i]
// (w = grad_out and X = q or k for the bias A2; w = attn and X = grad_out for the value term A4).
// The reference adds every (pair, i, ax) term with a global atomic into a 9216-float table
// (relative_pos_encoding_cuda_kernel_v2.cu:327-332, :478-480).  Here the sum is factored per row:
//     H_row[ax][r]   = sum of w over the row's pairs with rel == r        (a 3 x L histogram)
//     grad_table    += H_row^T (3L x 1)  x  X_row (1 x D)                 (an outer product)
// so per pair only 3 LDS adds remain (one lane per axis), and the outer products of four consecutive
// rows are one K=4 step of v_mfma_f32_16x16x4_f32 per 16-bin tile: D[bin, i] += A[bin, row] * B[row, i].
//
// The histogram is kept in 32-bit FIXED POINT with a per-row power-of-two scale: on gfx950 an LDS float
// atomic (ds_add_f32) retires ~1 lane per 3 cycles for the whole CU (192 cycles per wave instruction,
// tools/ubench/lds_atomic.hip), an LDS integer atomic (ds_add_u32) runs at LDS write speed (~9-20 cycles per
// wave instruction) - 10-40x faster, and integer sums do not depend on the order of the adds.  Per row:
//     E, n        max|w| < 2^E over the row's pairs, n pairs          (one extra sweep over w)
//     S           = 30 - bits(n) - E, so that |sum of any bin| * 2^S < 2^30
//     hist       += rne(w * 2^S)                                      (ds_add_u32)
//     B operand   = X_row * 2^-S  (exact), A operand = float(hist)     (exact below 2^24, else rne)
// Each term is rounded to max|w| * 2^-(29 - bits(n)), i.e. at n <= 64 pairs finer than the fp32 epsilon of the
// row's largest weight - the same size as the rounding of an fp32 running sum, without its order dependence.
// A row whose weights contain Inf/NaN poisons its tiles with NaN (the reference would propagate them too).
// Each wave keeps its whole table-gradient slice (HG heads x 3L x 16, 144 registers at L=64) in MFMA
// accumulators for the entire launch; the workgroup's four slices are merged in LDS once at the end and
// flushed with one global atomic per table entry and workgroup.  f32 MFMA is exact fp32 (fma chain).
//
// grad_q / grad_k / grad_attn are produced as in rpe.hip (table slices staged in LDS, no atomics).
// Used for D = 16 and L <= 80 (all shipped configs) when a CSC view is set; otherwise rpe.hip's kernels run.
#include "rpe_common.h"
#include <cstdlib>

namespace p2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int HG, int TA>
struct TableGrad {
    static constexpr int LP = TA * 16;                 // padded bins per axis
    static constexpr int ROW = HG * 3 * LP + 16;       // floats per k-row of the per-wave histogram (+16: the four
                                                       // k-rows read by one ds_read land on different banks)
    static constexpr int HIST = 4 * ROW;               // floats per wave
    static constexpr int XS = 4 * HG * 16;             // floats per wave: X rows of the current group
    f32x4 acc[HG][3 * TA];

    __device__ __forceinline__ void init(float *hist, float *xs, int lane) {
#pragma unroll
        for (int t = 0; t < HG; t++)
#pragma unroll
            for (int i = 0; i < 3 * TA; i++) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int x = lane; x < HIST; x += 64) hist[x] = 0.f;
        for (int x = lane; x < XS; x += 64) xs[x] = 0.f;
    }
    // fixed-point scale of a row: maxbits = bit pattern of max|w| over its n pairs
    struct Scale {
        float mul, inv;
    };
    static __device__ __forceinline__ Scale row_scale(unsigned maxbits, int n) {
        Scale sc;
        if (maxbits >= 0x7f800000u) {  // Inf / NaN among the weights
            sc.mul = 0.f;
            sc.inv = __uint_as_float(0x7fc00000u);
            return sc;
        }
        const int E = (int)(maxbits >> 23) - 126;   // max|w| < 2^E
        const int bits_n = 32 - __clz(max(n, 1));   // n < 2^bits_n
        const int S = max(-126, min(126, 30 - bits_n - E));
        sc.mul = __uint_as_float((unsigned)(S + 127) << 23);
        sc.inv = __uint_as_float((unsigned)(127 - S) << 23);
        return sc;
    }
    // lane c (< 3) of a pair's lane group owns axis c
    __device__ __forceinline__ void add(float *hist, int kk, int t, int c, int r, float w, float mul) {
        if (c < 3) atomicAdd(reinterpret_cast<int *>(&hist[kk * ROW + (t * 3 + c) * LP + r]), __float2int_rn(w * mul));
    }
    __device__ __forceinline__ void put_x(float *xs, int kk, int t, int c, float4 x4, float inv) {
        *reinterpret_cast<float4 *>(&xs[(kk * HG + t) * 16 + 4 * c]) = make_float4(x4.x * inv, x4.y * inv, x4.z * inv, x4.w * inv);
    }
    // consume the histograms of the (up to) four rows collected since the last call
    __device__ __forceinline__ void mma_group(float *hist, const float *xs, int lane) {
        const int kq = lane >> 4, col = lane & 15;
#pragma unroll
        for (int t = 0; t < HG; t++) {
            const float b = xs[(kq * HG + t) * 16 + col];
#pragma unroll
            for (int tile = 0; tile < 3 * TA; tile++) {
                int *hp = reinterpret_cast<int *>(&hist[kq * ROW + (t * 3 + tile / TA) * LP + (tile % TA) * 16 + col]);
                const float a = (float)*hp;
                *hp = 0;
                acc[t][tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t][tile], 0, 0, 0);
            }
        }
    }
    // C/D layout of 16x16x4: lane holds D[row = (lane>>4)*4 + reg][col = lane&15]  (row = bin, col = i)
    // The four waves add their slices into the (zeroed) image G one after the other with plain LDS
    // read-add-write - within a wave every (bin, i) is touched by exactly one lane - instead of 144 float
    // LDS atomics per wave (see the header: those serialize CU-wide).  Called by all waves of the workgroup.
    __device__ __forceinline__ void merge(float *G, int L, int hgn, int lane, int wave) {
        for (int turn = 0; turn < 4; turn++) {
            if (wave == turn) {
#pragma unroll
                for (int t = 0; t < HG; t++) {
                    if (t < hgn) {
#pragma unroll
                        for (int tile = 0; tile < 3 * TA; tile++) {
#pragma unroll
                            for (int reg = 0; reg < 4; reg++) {
                                const int bin = (tile % TA) * 16 + (lane >> 4) * 4 + reg;
                                const float v = acc[t][tile][reg];
                                if (bin < L && v != 0.f) G[((t * 3 + tile / TA) * L + bin) * 16 + (lane & 15)] += v;
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
};

// ------------------------------------------------------------------------------------------------
// A2 backward, one side:  rows = queries (CSR, pair_map == nullptr, X = q, out = grad_q stored)
//                         rows = keys    (CSC, pair_map = csc_pair,  X = k, out = grad_k accumulated)
// ------------------------------------------------------------------------------------------------
template <int HG, int TA, bool ACCUM_OUT>
__global__ __launch_bounds__(256, 2) void a2_bwd_side_mfma_kernel(int N, int h, int L, const float *__restrict__ go,
                                                                  const float *__restrict__ X, const int *__restrict__ offs,
                                                                  const int *__restrict__ pair_map, const float *__restrict__ table,
                                                                  const int *__restrict__ rel, float *__restrict__ grad_x,
                                                                  float *__restrict__ grad_table, int ablate) {
    constexpr int D = 16;
    using TG = TableGrad<HG, TA>;
    P2_WALK_PROLOGUE
    float *T = lds;                                     // [hgn][3][L][16]   (re-used as the merged gradient image at the end)
    float *hist = lds + HG * 3 * L * D + wave * TG::HIST;
    float *xs = lds + HG * 3 * L * D + 4 * TG::HIST + wave * TG::XS;
    stage_table<D>(T, table, L, h, h0, hgn);
    TG tg;
    tg.init(hist, xs, lane);
    __syncthreads();
    int kk = 0;
    for (int row = blockIdx.x * 4 + wave; row < N; row += gridDim.x * 4) {
        float4 x4[HG], acc[HG];
        const int s = offs[row], e = offs[row + 1];
        // largest |w| of the row -> its fixed-point scale (lane (p, c) looks at head c of pair slot p)
        unsigned mxb = 0u;
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int slot = m0 + p;
            if (slot < e && c < hgn) {
                const int m = pair_map ? pair_map[slot] : slot;
                mxb = max(mxb, __float_as_uint(fabsf(go[(size_t)m * h + h0 + c])));
            }
        }
        const typename TG::Scale sc = TG::row_scale(wave_max_u32(mxb), e - s);
#pragma unroll
        for (int t = 0; t < HG; t++) {
            x4[t] = t < hgn ? ldg4(X + (size_t)row * C + (h0 + t) * D + 4 * c) : make_float4(0, 0, 0, 0);
            acc[t] = make_float4(0, 0, 0, 0);
            if (p == 0) tg.put_x(xs, kk, t, c, x4[t], sc.inv);
        }
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int slot = m0 + p;
            if (slot < e) {
                const int m = pair_map ? pair_map[slot] : slot;
                const int r0 = clampr(rel[m * 3 + 0], L), r1 = clampr(rel[m * 3 + 1], L), r2 = clampr(rel[m * 3 + 2], L);
                const int rc = c == 0 ? r0 : (c == 1 ? r1 : r2);
#pragma unroll
                for (int t = 0; t < HG; t++) {
                    if (t < hgn) {
                        const float g = go[(size_t)m * h + h0 + t];
                        acc[t] = fma4(g, tsum<D>(T, L, t, r0, r1, r2, c), acc[t]);
                        if (!(ablate & 1)) tg.add(hist, kk, t, c, rc, g, sc.mul);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < HG; t++) {
            if (t < hgn) {
                float4 tot = xor_sum4<LPG, 64>(acc[t]);
                if (p == 0) {
                    float *o = grad_x + (size_t)row * C + (h0 + t) * D + 4 * c;
                    if (ACCUM_OUT) tot = add4(tot, ldg4(o));
                    stg4(o, tot);
                }
            }
        }
        if (++kk == 4) {
            if (!(ablate & 2)) tg.mma_group(hist, xs, lane);
            kk = 0;
        }
    }
    if (kk && !(ablate & 2)) tg.mma_group(hist, xs, lane);
    __syncthreads();                 // every wave is done reading T
    zero_lds<D>(T, tsz);
    __syncthreads();
    if (!(ablate & 4)) tg.merge(T, L, hgn, lane, wave);
    if (!(ablate & 8)) flush_table<D>(T, grad_table, L, h, h0, hgn);
}

// ------------------------------------------------------------------------------------------------
// A4 backward, by query: grad_attn[m,hh] = <Tv(m) + v[idx1[m]], grad_out[q]>;  grad_table from H(attn) x grad_out
// ------------------------------------------------------------------------------------------------
template <int HG, int TA>
__global__ __launch_bounds__(256, 2) void a4_bwd_query_mfma_kernel(int N, int h, int L, const float *__restrict__ go,
                                                                   const int *__restrict__ offs, const int *__restrict__ idx1,
                                                                   const float *__restrict__ attn, const float *__restrict__ v,
                                                                   const float *__restrict__ table, const int *__restrict__ rel,
                                                                   float *__restrict__ grad_attn, float *__restrict__ grad_table, int ablate) {
    constexpr int D = 16;
    using TG = TableGrad<HG, TA>;
    P2_WALK_PROLOGUE
    float *T = lds;
    float *hist = lds + HG * 3 * L * D + wave * TG::HIST;
    float *xs = lds + HG * 3 * L * D + 4 * TG::HIST + wave * TG::XS;
    stage_table<D>(T, table, L, h, h0, hgn);
    TG tg;
    tg.init(hist, xs, lane);
    __syncthreads();
    int kk = 0;
    for (int qi = blockIdx.x * 4 + wave; qi < N; qi += gridDim.x * 4) {
        float4 g4[HG];
        const int s = offs[qi], e = offs[qi + 1];
        unsigned mxb = 0u;
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int m = m0 + p;
            if (m < e && c < hgn) mxb = max(mxb, __float_as_uint(fabsf(attn[(size_t)m * h + h0 + c])));
        }
        const typename TG::Scale sc = TG::row_scale(wave_max_u32(mxb), e - s);
#pragma unroll
        for (int t = 0; t < HG; t++) {
            g4[t] = t < hgn ? ldg4(go + (size_t)qi * C + (h0 + t) * D + 4 * c) : make_float4(0, 0, 0, 0);
            if (p == 0) tg.put_x(xs, kk, t, c, g4[t], sc.inv);
        }
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int m = m0 + p;
            const bool valid = m < e;
            const int mm = valid ? m : s;
            const int j = idx1[mm];
            const int r0 = clampr(rel[mm * 3 + 0], L), r1 = clampr(rel[mm * 3 + 1], L), r2 = clampr(rel[mm * 3 + 2], L);
            const int rc = c == 0 ? r0 : (c == 1 ? r1 : r2);
            float keep = 0.f;
#pragma unroll
            for (int t = 0; t < HG; t++) {
                if (t < hgn) {
                    const float4 v4 = ldg4(v + (size_t)j * C + (h0 + t) * D + 4 * c);
                    float part = dot4(add4(tsum<D>(T, L, t, r0, r1, r2, c), v4), g4[t]);
                    float tot = xor_sum<1, LPG>(part);
                    if (c == t) keep = tot;
                    if (valid && !(ablate & 1)) tg.add(hist, kk, t, c, rc, attn[(size_t)m * h + h0 + t], sc.mul);
                }
            }
            if (valid && c < hgn) grad_attn[(size_t)m * h + h0 + c] = keep;
        }
        if (++kk == 4) {
            if (!(ablate & 2)) tg.mma_group(hist, xs, lane);
            kk = 0;
        }
    }
    if (kk && !(ablate & 2)) tg.mma_group(hist, xs, lane);
    __syncthreads();
    zero_lds<D>(T, tsz);
    __syncthreads();
    if (!(ablate & 4)) tg.merge(T, L, hgn, lane, wave);
    if (!(ablate & 8)) flush_table<D>(T, grad_table, L, h, h0, hgn);
}

template <int HG, int TA>
static size_t mfma_lds_bytes(int L) {
    using TG = TableGrad<HG, TA>;
    return ((size_t)HG * 3 * L * 16 + 4 * TG::HIST + 4 * TG::XS) * sizeof(float);
}

// diagnostic only (tools/bench_ops.py): P2_ABLATE bit 0 = no histogram adds, 1 = no MFMA, 2 = no merge, 3 = no flush
static int ablate_mask() {
    static const int v = getenv("P2_ABLATE") ? atoi(getenv("P2_ABLATE")) : 0;
    return v;
}

static int mfma_blocks(int rows, int groups) {
    int want = div_up(rows, 4);
    int cap = kNumCU * 2;
    if (groups > 1) cap = max(kNumCU * 2 / groups, kNumCU / 2);
    return min(want, cap);
}

template <int TA>
static bool launch_a2(int N, int NK, int h, int L, const float *go, const float *q, const int *offs, const float *k,
                      const float *table_q, const float *table_k, const int *rel, const int *co, const int *cp,
                      float *grad_q, float *grad_k, float *gtq, float *gtk, hipStream_t st) {
    // heads per workgroup: 3 at h >= 3 (75 KB LDS at L=64 -> two workgroups per CU)
    auto go_hg = [&](auto tag) {
        constexpr int HG = decltype(tag)::value;
        const int groups = div_up(h, HG);
        const size_t lds = mfma_lds_bytes<HG, TA>(L);
        allow_big_lds(a2_bwd_side_mfma_kernel<HG, TA, false>, lds);
        allow_big_lds(a2_bwd_side_mfma_kernel<HG, TA, true>, lds);
        hipLaunchKernelGGL((a2_bwd_side_mfma_kernel<HG, TA, false>), dim3(mfma_blocks(N, groups), groups), dim3(256), lds, st,
                           N, h, L, go, q, offs, (const int *)nullptr, table_q, rel, grad_q, gtq, ablate_mask());
        hipLaunchKernelGGL((a2_bwd_side_mfma_kernel<HG, TA, true>), dim3(mfma_blocks(NK, groups), groups), dim3(256), lds, st,
                           NK, h, L, go, k, co, cp, table_k, rel, grad_k, gtk, ablate_mask());
    };
    if (h >= 3) go_hg(std::integral_constant<int, 3>{});
    else if (h == 2) go_hg(std::integral_constant<int, 2>{});
    else go_hg(std::integral_constant<int, 1>{});
    return true;
}

template <int TA>
static bool launch_a4(int N, int h, int L, const float *go, const int *offs, const int *idx1, const float *attn, const float *v,
                      const float *table, const int *rel, float *grad_attn, float *grad_table, hipStream_t st) {
    auto go_hg = [&](auto tag) {
        constexpr int HG = decltype(tag)::value;
        const int groups = div_up(h, HG);
        const size_t lds = mfma_lds_bytes<HG, TA>(L);
        allow_big_lds(a4_bwd_query_mfma_kernel<HG, TA>, lds);
        hipLaunchKernelGGL((a4_bwd_query_mfma_kernel<HG, TA>), dim3(mfma_blocks(N, groups), groups), dim3(256), lds, st,
                           N, h, L, go, offs, idx1, attn, v, table, rel, grad_attn, grad_table, ablate_mask());
    };
    if (h >= 3) go_hg(std::integral_constant<int, 3>{});
    else if (h == 2) go_hg(std::integral_constant<int, 2>{});
    else go_hg(std::integral_constant<int, 1>{});
    return true;
}

// entry points used by rpe.hip; return false when the shape is outside this file's fast path
bool a2_bwd_mfma(int N, int NK, int h, int hdim, int L, const float *go, const float *q, const int *offs, const float *k,
                 const float *table_q, const float *table_k, const int *rel, const int *co, const int *cp,
                 float *grad_q, float *grad_k, float *gtq, float *gtk) {
    if (hdim != 16 || co == nullptr || L < 1 || L > 80) return false;
    hipStream_t st = state().stream;
    if (L <= 64) return launch_a2<4>(N, NK, h, L, go, q, offs, k, table_q, table_k, rel, co, cp, grad_q, grad_k, gtq, gtk, st);
    return launch_a2<5>(N, NK, h, L, go, q, offs, k, table_q, table_k, rel, co, cp, grad_q, grad_k, gtq, gtk, st);
}

bool a4_bwd_mfma(int N, int h, int hdim, int L, const float *go, const int *offs, const int *idx1, const float *attn,
                 const float *v, const float *table, const int *rel, float *grad_attn, float *grad_table) {
    if (hdim != 16 || L < 1 || L > 80) return false;
    hipStream_t st = state().stream;
    if (L <= 64) return launch_a4<4>(N, h, L, go, offs, idx1, attn, v, table, rel, grad_attn, grad_table, st);
    return launch_a4<5>(N, h, L, go, offs, idx1, attn, v, table, rel, grad_attn, grad_table, st);
}

}  // namespace p2
