// Backward of A2 / A4 for D = 16 with a key-major (CSC) view of the pair list, gfx950: no global atomics
// except the final table flush, table gradients on the matrix cores.
//
// The walk over a row's pairs (pair -> rel index -> LDS table rows) is a chain of dependent accesses and a
// row is only ~3 wave passes long, so these kernels are latency-bound: time goes with 1/(waves per CU)
// (rpe.hip, A2 forward).  A kernel that keeps a table-gradient slice in MFMA accumulators per wave
// (144 registers) runs at 8 waves per CU and spent 75 % of its time in the walk it shares with the row
// gradients.  So the work is split by what limits it:
//
//   rows_table_sum_kernel   grad_q / grad_k rows:  sum over the row's pairs of w * T(rel)     32 waves / CU
//   a4_bwd_attn_kernel      grad_attn per pair:    <T(rel) + v[key], grad_out[query]>          32 waves / CU
//   table_grad_kernel       all three table gradients (below)                                  24 waves / CU
//
// Table gradient.  grad_table[r, hh, i, ax] = sum over pairs m with rel[m, ax] == r of w[m, hh] * X[row(m), hh, i]
// (w = grad_out, X = q or k for the bias A2; w = attn, X = grad_out for the value term A4).  The reference
// adds every (pair, i, ax) term with a global atomic into a 9216-float table
// (relative_pos_encoding_cuda_kernel_v2.cu:327-332, :478-480).  Here the sum is factored per row:
//     H_row[ax][r]   = sum of w over the row's pairs with rel == r        (a 3 x L histogram per head)
//     grad_table    += H_row^T (3L x 1)  x  X_row (1 x 16)                (an outer product)
// A workgroup of 12 waves takes 12 rows at a time: every wave builds the histogram of one row in LDS
// (3 integer LDS adds per pair and head), then - after a barrier - the 12 outer products are three K=4
// steps of v_mfma_f32_16x16x4_f32 per 16-bin tile, D[bin, i] += A[bin, row] * B[row, i], and the tiles of
// the table slice (HG heads x 3 axes x L/16) are dealt out over the 12 waves: 3 tiles = 12 accumulator
// registers per wave instead of the whole slice.  Nothing has to be merged at the end; every wave flushes
// its own tiles with one global atomic per table entry.  f32 MFMA is exact fp32 (fma chain).
//
// The histogram is kept in 32-bit FIXED POINT with a per-row power-of-two scale: on gfx950 an LDS float
// atomic (ds_add_f32) retires ~1 lane per 3 cycles for the whole CU (192 cycles per wave instruction,
// tools/ubench/lds_atomic.hip), an LDS integer atomic (ds_add_u32) runs at LDS write speed (~9-20 cycles per
// wave instruction), and integer sums do not depend on the order of the adds.  Per row:
//     E, n        max|w| < 2^E over the row's pairs, n pairs          (one extra sweep over w)
//     S           = 30 - bits(n) - E, so that |sum of any bin| * 2^S < 2^30
//     hist       += rne(w * 2^S)                                      (ds_add_u32)
//     B operand   = X_row * 2^-S  (exact), A operand = float(hist)     (exact below 2^24, else rne)
// Each term is rounded to max|w| * 2^-(29 - bits(n)), i.e. at n <= 64 pairs finer than the fp32 epsilon of the
// row's largest weight - the same size as the rounding of an fp32 running sum, without its order dependence.
// A row whose weights contain Inf/NaN poisons its tiles with NaN (the reference would propagate them too).
//
// Used for D = 16 and L <= 80 (all shipped configs) when a CSC view is set; otherwise rpe.hip's kernels run.
#include "rpe_common.h"
#include <cstdio>
#include <cstdlib>

namespace p2 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// ------------------------------------------------------------------------------------------------
// grad_x[row, hh, :] (= | +=) sum over the row's pairs of w[m, hh] * (T[r0,.,0] + T[r1,.,1] + T[r2,.,2])
//   rows = queries (CSR, pair_map == nullptr, stored)   |   rows = keys (CSC, pair_map = csc_pair, accumulated)
// ------------------------------------------------------------------------------------------------
template <int HG, bool ACCUM_OUT>
__global__ __launch_bounds__(512, 8) void rows_table_sum_kernel(int N, int h, int L, const float *__restrict__ w,
                                                                const int *__restrict__ offs, const int *__restrict__ pair_map,
                                                                const float *__restrict__ table, const int *__restrict__ rel,
                                                                float *__restrict__ grad_x, const int *__restrict__ rord) {
    constexpr int D = 16;
    P2_WALK_PROLOGUE
    float *T = lds;
    stage_table<D>(T, table, L, h, h0, hgn);
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    for (RowSlots slots(rord, N, wpb, wave); slots.more(); slots.next()) {
        const int row = slots.row();
        float4 acc[HG];
#pragma unroll
        for (int t = 0; t < HG; t++) acc[t] = make_float4(0, 0, 0, 0);
        const int s = offs[row], e = offs[row + 1];
        float4 prev[HG];  // ACCUM_OUT: the row's running sum, requested now and consumed after the walk
        if (ACCUM_OUT) {
#pragma unroll
            for (int t = 0; t < HG; t++) prev[t] = ldg4(grad_x + (size_t)row * C + (h0 + min(t, hgn - 1)) * D + 4 * c);
        }
        // by key (ACCUM_OUT) the pair id of the next pass is requested together with the weights of this one
        int mn = ACCUM_OUT ? pair_map[max(0, min(s + p, e - 1))] : 0;
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int slot = m0 + p;
            const int m = ACCUM_OUT ? mn : slot;
            if (ACCUM_OUT) mn = pair_map[min(slot + PPW, e - 1)];
            if (slot < e) {
                // no per-head guards (rpe.hip, a2_fwd_kernel): a slot past the group's last head repeats it.  The
                // scheduling fence keeps the clamps (the first use of rel) behind the weight loads: one round trip.
                const int q0 = rel[m * 3 + 0], q1 = rel[m * 3 + 1], q2 = rel[m * 3 + 2];
                float g[HG];
#pragma unroll
                for (int t = 0; t < HG; t++) g[t] = w[(size_t)m * h + h0 + min(t, hgn - 1)];
                __builtin_amdgcn_sched_barrier(0);
                const int r0 = clampr(q0, L), r1 = clampr(q1, L), r2 = clampr(q2, L);
#pragma unroll
                for (int t = 0; t < HG; t++) acc[t] = fma4(g[t], tsum<D>(T, L, min(t, hgn - 1), r0, r1, r2, c), acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < HG; t++) {
            if (t < hgn) {
                float4 tot = xor_sum4<LPG, 64>(acc[t]);
                if (p == 0) {
                    float *o = grad_x + (size_t)row * C + (h0 + t) * D + 4 * c;
                    if (ACCUM_OUT) tot = add4(tot, prev[t]);
                    stg4(o, tot);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// A4 backward, per pair: grad_attn[m, hh] = <Tv(m, hh, :) + v[idx1[m], hh, :], grad_out[query, hh, :]>
// ------------------------------------------------------------------------------------------------
template <int HG>
__global__ __launch_bounds__(512, 8) void a4_bwd_attn_kernel(int N, int h, int L, const float *__restrict__ go,
                                                             const int *__restrict__ offs, const int *__restrict__ idx1,
                                                             const float *__restrict__ v, const float *__restrict__ table,
                                                             const int *__restrict__ rel, float *__restrict__ grad_attn, const int *__restrict__ rord) {
    constexpr int D = 16;
    P2_WALK_PROLOGUE
    float *T = lds;
    stage_table<D>(T, table, L, h, h0, hgn);
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    for (RowSlots slots(rord, N, wpb, wave); slots.more(); slots.next()) {
        const int qi = slots.row();
        float4 g4[HG];
#pragma unroll
        for (int t = 0; t < HG; t++)
            g4[t] = t < hgn ? ldg4(go + (size_t)qi * C + (h0 + t) * D + 4 * c) : make_float4(0, 0, 0, 0);
        const int s = offs[qi], e = offs[qi + 1];
        // the ids of the next pass are requested together with the value rows of this one: one round trip per pass
        PairIds nx = load_pair_ids(idx1, rel, max(0, min(s + p, e - 1)));
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int m = m0 + p;
            const bool valid = m < e;
            const PairIds cur = nx;
            nx = load_pair_ids(idx1, rel, min(m + PPW, e - 1));
            const int j = cur.j;
            float keep = 0.f;
            float4 v4[HG];
#pragma unroll
            for (int t = 0; t < HG; t++) v4[t] = ldg4(v + (size_t)j * C + (h0 + min(t, hgn - 1)) * D + 4 * c);
            __builtin_amdgcn_sched_barrier(0);
            const int r0 = clampr(cur.q0, L), r1 = clampr(cur.q1, L), r2 = clampr(cur.q2, L);
#pragma unroll
            for (int t = 0; t < HG; t++) {
                float part = dot4(add4(tsum<D>(T, L, min(t, hgn - 1), r0, r1, r2, c), v4[t]), g4[t]);
                float tot = xor_sum<1, LPG>(part);
                if (c == t) keep = tot;
            }
            if (valid && c < hgn) grad_attn[(size_t)m * h + h0 + c] = keep;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// table gradient: grad_table[r, hh, i, ax] += sum over rows, over the row's pairs with rel[m, ax] == r, of
//                 w[m, hh] * X[row, hh, i]
// ------------------------------------------------------------------------------------------------
struct FixScale {
    float mul, inv;
};
// fixed-point scale of a row: maxbits = bit pattern of max|w| over its n pairs
__device__ __forceinline__ FixScale row_scale(unsigned maxbits, int n) {
    FixScale sc;
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

// value of lane t (0..2) of each quad in all four lanes of the quad (DPP quad_perm, all lanes must be active)
__device__ __forceinline__ float quad_bcast(float v, int t) {
    const int x = __float_as_int(v);
    int r;
    if (t == 0) r = __builtin_amdgcn_update_dpp(0, x, 0x00, 0xF, 0xF, false);
    else if (t == 1) r = __builtin_amdgcn_update_dpp(0, x, 0x55, 0xF, 0xF, false);
    else r = __builtin_amdgcn_update_dpp(0, x, 0xAA, 0xF, 0xF, false);
    return __int_as_float(r);
}

constexpr int TG_WAVES = 12;  // rows per group = K of the outer-product step (3 x 4)
constexpr int TG_MAXP = 8;    // passes (of 16 pairs) a row may have to be walked from registers

template <int HG, int TA>
struct TableGeo {
    static constexpr int LP = TA * 16;                 // padded bins per axis
    static constexpr int ROW = HG * 3 * LP + 16;       // ints per histogram row (+16: the four k-rows read by one
                                                       // ds_read land on different banks)
    static constexpr int GROUPS = HG * TA;             // (head, 16-bin tile); a wave owns whole groups = 3 axis tiles each
    static constexpr int GPW = (GROUPS + TG_WAVES - 1) / TG_WAVES;
    static constexpr size_t walk_bytes() { return (size_t)TG_WAVES * (ROW + HG * 16) * 4; }
    static constexpr size_t flush_bytes() { return (size_t)TG_WAVES * 16 * 48 * 4; }  // one group per wave at a time
    static constexpr size_t lds_bytes() { return walk_bytes() > flush_bytes() ? walk_bytes() : flush_bytes(); }
};

// STAMP: diagnostic build only (P2_TG_STAMPS=1): cycle sums of the phases of every wave of workgroup 0 -> dbg
template <int HG, int TA, bool STAMP = false>
__global__ __launch_bounds__(TG_WAVES * 64, 6) void table_grad_kernel(int N, int h, int L, const float *__restrict__ w,
                                                                      const float *__restrict__ X, const int *__restrict__ offs,
                                                                      const int *__restrict__ pair_map, const int *__restrict__ rel,
                                                                      float *__restrict__ grad_table,
                                                                      const int *__restrict__ rord, unsigned long long *__restrict__ dbg = nullptr) {
    unsigned long long c_ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    auto stamp = [&](int ph) {  // everything issued so far has completed; the time since the last stamp goes to phase ph
        if (STAMP) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            c_ph[ph] += t - t_last;
            t_last = t;
        }
    };
    if (STAMP) t_last = __builtin_amdgcn_s_memtime();
    constexpr int D = 16, NW = TG_WAVES;
    using G = TableGeo<HG, TA>;
    extern __shared__ float lds[];
    int *hist = reinterpret_cast<int *>(lds);        // [NW][ROW]
    float *xs = lds + NW * G::ROW;                    // [NW][HG][16]: X rows of the group, scaled by 2^-S
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int C = h * D;
    const int p = lane >> 2, c = lane & 3;            // pair slot of a pass; axis (c < 3) / head (c < hgn) of this lane
    const int h0 = blockIdx.y * HG;
    const int hgn = min(HG, h - h0);
    const int kq = lane >> 4, col = lane & 15;

    f32x4 acc[G::GPW][3];  // [owned group][axis]
#pragma unroll
    for (int i = 0; i < G::GPW; i++)
#pragma unroll
        for (int ax = 0; ax < 3; ax++) acc[i][ax] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int x = threadIdx.x; x < NW * G::ROW; x += NW * 64) hist[x] = 0;
    __syncthreads();

    int *myh = hist + wave * G::ROW;
    // Rows are dealt out dynamically inside the workgroup's contiguous share [rb, re), and a row longer than
    // TG_MAXP passes is taken in segments of 16 * TG_MAXP pairs (the gradient is linear in the pairs, a segment
    // is simply one more outer product with the same X row).  Key-major rows are very uneven - a stratified key
    // is shared by several times more queries than an ordinary one - and every group ends in a barrier.
    __shared__ int next_row;
    const int per = (N + gridDim.x - 1) / gridDim.x;
    // (rows in window order, common.h: the shares are runs of the order, consecutive runs on one XCD)
    const int share = (rord != nullptr && (gridDim.x & 7) == 0) ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
    const int rb = min(N, share * per), re = min(N, rb + per);
    if (threadIdx.x == 0) next_row = rb;
    __syncthreads();
    stamp(0);  // 0: prologue (zeroing the histograms)
    // A wave claims its next row, and requests that row's bounds, while it still works on the last segment of the
    // current one: the claim (an LDS round trip) and the bounds (a memory round trip) are off the dependent chain
    // of the segment that uses them.
    auto claim = [&]() -> int {
        int r = 0;
        if (lane == 0) r = atomicAdd(&next_row, 1);
        r = __builtin_amdgcn_readfirstlane(r);
        return r < re ? (rord != nullptr ? rord[r] : r) : -1;
    };
    int row = -1, cur = 0, end = 0;
    int nrow = claim(), ncur = 0, nend = 0;
    if (nrow >= 0) {
        ncur = offs[nrow];
        nend = offs[nrow + 1];
    }
    for (;;) {
        if (cur >= end) {  // this wave's row is finished: take the claimed one (none left: row = -1 from here on)
            row = nrow;
            cur = row >= 0 ? ncur : 0;
            end = row >= 0 ? nend : 0;
            nrow = -1;
        }
        const bool last_segment = row >= 0 && end - cur <= 16 * TG_MAXP;
        if (row >= 0) {
            const int s = cur, e = min(end, cur + 16 * TG_MAXP);
            cur = e;
            const int np = (e - s + 15) >> 4;  // wave passes of 16 pairs
            // The loads of ALL passes are issued before anything depends on them - pair id, then rel index
            // (lane c < 3: axis c) and weight (lane c < hgn: head c) - and stay in registers: one dependent
            // chain per segment instead of one per pass, and the sweep for the fixed-point scale re-reads nothing.
            // Every load is UNCONDITIONAL, with a clamped index (a slot past the segment's end repeats its last pair, an
            // idle lane reads its neighbour's value), and nothing consumes a loaded value before the last load is issued:
            // a predicated load sits in a block of its own that ends with a wait for it, and the running max below,
            // when it was folded into the load loop, made the weights eight dependent round trips.
            int mreg[TG_MAXP], rreg[TG_MAXP];
            float wreg[TG_MAXP];
#pragma unroll
            for (int i = 0; i < TG_MAXP; i++) { mreg[i] = -1; rreg[i] = 0; wreg[i] = 0.f; }
            // the row of X: every lane loads (lanes >= 4*HG repeat head 0's quarters), only lanes < 4*HG store it below
            const float4 x4 = ldg4(X + (size_t)row * C + (h0 + min(lane >> 2, hgn - 1) % HG) * D + 4 * c);
            if (e > s) {  // wave-uniform (an empty row has nothing to read, and s may be one past the last pair)
#pragma unroll
                for (int i = 0; i < TG_MAXP; i++) {
                    const int slot = min(s + i * 16 + p, e - 1);
                    mreg[i] = pair_map ? pair_map[slot] : slot;
                }
            }
            stamp(1);  // 1: row bounds + pair ids
            if (last_segment) {  // wave-uniform; behind the pair ids so that the claim's LDS round trip overlaps them
                nrow = claim();
                if (nrow >= 0) {
                    ncur = offs[nrow];
                    nend = offs[nrow + 1];
                }
            }
            if (e > s) {
#pragma unroll
                for (int i = 0; i < TG_MAXP; i++) {
                    rreg[i] = rel[mreg[i] * 3 + min(c, 2)];
                    wreg[i] = w[(size_t)mreg[i] * h + h0 + min(c, hgn - 1)];
                }
            }
            stamp(2);  // 2: claim of the next row + rel indices, weights, X row
            unsigned mxb = 0u;
#pragma unroll
            for (int i = 0; i < TG_MAXP; i++) {
                const bool live = s + i * 16 + p < e;
                if (!live) mreg[i] = -1;
                wreg[i] = (live && c < hgn) ? wreg[i] : 0.f;
                mxb = max(mxb, __float_as_uint(fabsf(wreg[i])));
            }
            const FixScale sc = row_scale(wave_max_u32(mxb), e - s);
            if (lane < HG * 4)  // lane = t*4 + quarter: the X row (loaded with the batch above), scaled by 2^-S
                *reinterpret_cast<float4 *>(&xs[(wave * HG + (lane >> 2)) * 16 + 4 * c]) =
                    make_float4(x4.x * sc.inv, x4.y * sc.inv, x4.z * sc.inv, x4.w * sc.inv);
#pragma unroll
            for (int i = 0; i < TG_MAXP; i++) {
                if (i < np) {  // wave-uniform
                    const int r = clampr(rreg[i], L);
#pragma unroll
                    for (int t = 0; t < HG; t++) {
                        const float wt = quad_bcast(wreg[i], t);  // weight of head t of this lane's pair
                        if (t < hgn && mreg[i] >= 0 && c < 3) atomicAdd(&myh[(t * 3 + c) * G::LP + r], __float2int_rn(wt * sc.mul));
                    }
                }
            }
        } else if (lane < HG * 4) {
            *reinterpret_cast<float4 *>(&xs[(wave * HG + (lane >> 2)) * 16 + 4 * c]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        stamp(3);  // 3: scale sweep + LDS atomics
        // the 12 histograms and X rows of the group are complete; stop when no wave had a segment
        const int more = __syncthreads_or(row >= 0);
        stamp(4);  // 4: waiting for the group's slowest wave
        if (!more) break;
#pragma unroll
        for (int i = 0; i < G::GPW; i++) {
            const int g = wave + i * NW;  // group = (head t, bin tile bt)
            if (g < G::GROUPS) {
                const int t = g / TA, bt = g % TA;
#pragma unroll
                for (int ks = 0; ks < NW / 4; ks++) {
                    const int rk = ks * 4 + kq;
                    const float b = xs[(rk * HG + t) * 16 + col];
#pragma unroll
                    for (int ax = 0; ax < 3; ax++) {
                        int *hp = &hist[rk * G::ROW + (t * 3 + ax) * G::LP + bt * 16 + col];
                        const float a = (float)*hp;
                        *hp = 0;
                        acc[i][ax] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][ax], 0, 0, 0);
                    }
                }
            }
        }
        stamp(5);  // 5: outer products
        __syncthreads();  // histograms are zero again, xs may be overwritten
        stamp(6);  // 6: second barrier
    }
    // Flush.  C/D layout of 16x16x4: lane holds D[row = (lane>>4)*4 + reg][col = lane&15]  (row = bin, col = i).
    // In the [L, h, 16, 3] table the 48 floats of one (bin, head) are contiguous, and a wave owns all three axis
    // tiles of its groups: the group goes through LDS in table order and leaves as 12 wave instructions of 64
    // consecutive floats (global float atomics run at full rate only for contiguous 128-256 B segments,
    // MI355X_MICROARCH.md; one dword per lane with a 12-byte stride was 3-5x slower).
    __syncthreads();  // the walk's LDS is free
    float *stage = lds + wave * (16 * 48);
#pragma unroll
    for (int i = 0; i < G::GPW; i++) {
        const int g = wave + i * NW;
        if (g < G::GROUPS) {
            const int t = g / TA, bt = g % TA;
            if (t < hgn) {
#pragma unroll
                for (int ax = 0; ax < 3; ax++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) stage[((kq * 4 + reg) * 16 + col) * 3 + ax] = acc[i][ax][reg];
                __builtin_amdgcn_s_waitcnt(0xC07F);  // this wave's LDS writes have landed (the region is private to the wave)
#pragma unroll
                for (int x = lane; x < 16 * 48; x += 64) {
                    const int bin = bt * 16 + x / 48;
                    const float v = stage[x];
                    if (bin < L && v != 0.f) atomicAdd(grad_table + ((size_t)bin * h + (h0 + t)) * 48 + x % 48, v);
                }
                __builtin_amdgcn_s_waitcnt(0xC07F);  // reads done before the next group overwrites the stage
            }
        }
    }
    stamp(7);  // 7: flush
    if (STAMP && dbg && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) dbg[wave * 8 + i] = c_ph[i];
    }
}

// ------------------------------------------------------------------------------------------------
// Fused backward of the whole window attention (optional module).  Two walks instead of seven:
//   wattn_bwd_query_kernel  per query:  grad_attn = <Tv + v[j], grad_out>,  softmax backward over the row,
//                           grad_logit stored, grad_q = sum grad_logit * k[j]  +  sum grad_logit * Tq
//   wattn_bwd_key_kernel    per key (CSC):  grad_k = sum grad_logit * q[i] + sum grad_logit * Tk,
//                           grad_v = sum attn * grad_out[i]
// plus the three table_grad launches.  Each sum is taken as the separate operators take it (same pair order,
// same butterflies, the two parts of grad_q / grad_k reduced separately and added at the end).
// ------------------------------------------------------------------------------------------------
constexpr int WB_MAXP = 8;

template <int HG>
__global__ __launch_bounds__(512, 4) void wattn_bwd_query_kernel(int N, int h, int L, const float *__restrict__ go,
                                                                 const int *__restrict__ offs, const int *__restrict__ idx1,
                                                                 const float *__restrict__ attn, const float *__restrict__ v,
                                                                 const float *__restrict__ k, const float *__restrict__ table_v,
                                                                 const float *__restrict__ table_q, const int *__restrict__ rel,
                                                                 float *__restrict__ grad_logit, float *__restrict__ grad_q, const int *__restrict__ rord) {
    constexpr int D = 16;
    P2_WALK_PROLOGUE
    float *Tv = lds, *Tq = lds + tsz;
    stage_table<D>(Tv, table_v, L, h, h0, hgn);
    stage_table<D>(Tq, table_q, L, h, h0, hgn);
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    auto slots_sum = [&](float x) {
        for (int st = LPG; st < 64; st <<= 1) x += __shfl_xor(x, st, 64);
        return x;
    };
    const bool mine = c < hgn;
    for (RowSlots slots(rord, N, wpb, wave); slots.more(); slots.next()) {
        const int qi = slots.row();
        float4 g4[HG], acc1[HG], acc2[HG];
#pragma unroll
        for (int t = 0; t < HG; t++) {
            g4[t] = t < hgn ? ldg4(go + (size_t)qi * C + (h0 + t) * D + 4 * c) : make_float4(0, 0, 0, 0);
            acc1[t] = acc2[t] = make_float4(0, 0, 0, 0);
        }
        const int s = offs[qi], e = offs[qi + 1];
        const int np = (e - s + PPW - 1) / PPW;
        // grad_attn of head c (kept by lane c of the pair's lane group) for pair m
        auto dattn = [&](int m, bool valid) -> float {
            const int mm = valid ? m : s;
            const int j = idx1[mm];
            const int r0 = clampr(rel[mm * 3 + 0], L), r1 = clampr(rel[mm * 3 + 1], L), r2 = clampr(rel[mm * 3 + 2], L);
            float keep = 0.f;
            float4 v4[HG];
#pragma unroll
            for (int t = 0; t < HG; t++) v4[t] = ldg4(v + (size_t)j * C + (h0 + min(t, hgn - 1)) * D + 4 * c);  // no per-head guards
#pragma unroll
            for (int t = 0; t < HG; t++) {
                const float tot = xor_sum<1, LPG>(dot4(add4(tsum<D>(Tv, L, min(t, hgn - 1), r0, r1, r2, c), v4[t]), g4[t]));
                if (c == t) keep = tot;
            }
            return keep;
        };
        // grad_q contributions of pair m, whose grad_logit of head c sits in lane c of its lane group
        auto accumulate = [&](int m, bool valid, float dl) {
            const int mm = valid ? m : s;
            const int j = idx1[mm];
            const int r0 = clampr(rel[mm * 3 + 0], L), r1 = clampr(rel[mm * 3 + 1], L), r2 = clampr(rel[mm * 3 + 2], L);
            float4 k4[HG];
#pragma unroll
            for (int t = 0; t < HG; t++) k4[t] = ldg4(k + (size_t)j * C + (h0 + min(t, hgn - 1)) * D + 4 * c);
#pragma unroll
            for (int t = 0; t < HG; t++) {
                const float w = valid ? quad_bcast(dl, t) : 0.f;  // a lane past the row's end adds zero
                acc1[t] = fma4(w, k4[t], acc1[t]);
                acc2[t] = fma4(w, tsum<D>(Tq, L, min(t, hgn - 1), r0, r1, r2, c), acc2[t]);
            }
        };
        if (e > s && np <= WB_MAXP) {
            float da[WB_MAXP], a[WB_MAXP];
            float dot = 0.f;
#pragma unroll
            for (int i = 0; i < WB_MAXP; i++) {
                da[i] = a[i] = 0.f;
                if (i < np) {  // wave-uniform
                    const int m = s + i * PPW + p;
                    const float x = dattn(m, m < e);
                    if (m < e && mine) {
                        da[i] = x;
                        a[i] = attn[(size_t)m * h + h0 + c];
                        dot = fmaf(a[i], da[i], dot);
                    }
                }
            }
            dot = slots_sum(dot);
#pragma unroll
            for (int i = 0; i < WB_MAXP; i++) {
                if (i < np) {
                    const int m = s + i * PPW + p;
                    const float dl = a[i] * (da[i] - dot);
                    if (m < e && mine) grad_logit[(size_t)m * h + h0 + c] = dl;
                    accumulate(m, m < e, dl);
                }
            }
        } else if (e > s) {
            float dot = 0.f;
            for (int m0 = s; m0 < e; m0 += PPW) {
                const int m = m0 + p;
                const float x = dattn(m, m < e);
                if (m < e && mine) {
                    grad_logit[(size_t)m * h + h0 + c] = x;  // parked: this very lane reads it back below
                    dot = fmaf(attn[(size_t)m * h + h0 + c], x, dot);
                }
            }
            dot = slots_sum(dot);
            for (int m0 = s; m0 < e; m0 += PPW) {
                const int m = m0 + p;
                float dl = 0.f;
                if (m < e && mine) {
                    dl = attn[(size_t)m * h + h0 + c] * (grad_logit[(size_t)m * h + h0 + c] - dot);
                    grad_logit[(size_t)m * h + h0 + c] = dl;
                }
                accumulate(m, m < e, dl);
            }
        }
#pragma unroll
        for (int t = 0; t < HG; t++) {
            if (t < hgn) {
                const float4 t1 = xor_sum4<LPG, 64>(acc1[t]), t2 = xor_sum4<LPG, 64>(acc2[t]);
                if (p == 0) stg4(grad_q + (size_t)qi * C + (h0 + t) * D + 4 * c, add4(t1, t2));
            }
        }
    }
}

template <int HG>
__global__ __launch_bounds__(512, 4) void wattn_bwd_key_kernel(int NK, int h, int L, const float *__restrict__ grad_logit,
                                                               const float *__restrict__ attn, const float *__restrict__ go,
                                                               const float *__restrict__ q, const int *__restrict__ csc_offs,
                                                               const int *__restrict__ csc_pair, const int *__restrict__ csc_query,
                                                               const float *__restrict__ table_k, const int *__restrict__ rel,
                                                               float *__restrict__ grad_k, float *__restrict__ grad_v, const int *__restrict__ rord) {
    constexpr int D = 16;
    P2_WALK_PROLOGUE
    float *Tk = lds;
    stage_table<D>(Tk, table_k, L, h, h0, hgn);
    __syncthreads();
    const int wpb = blockDim.x >> 6;
    for (RowSlots slots(rord, NK, wpb, wave); slots.more(); slots.next()) {
        const int kj = slots.row();
        float4 a1[HG], a2[HG], av[HG];
#pragma unroll
        for (int t = 0; t < HG; t++) a1[t] = a2[t] = av[t] = make_float4(0, 0, 0, 0);
        const int s = csc_offs[kj], e = csc_offs[kj + 1];
        for (int m0 = s; m0 < e; m0 += PPW) {
            const int slot = m0 + p;
            if (slot < e) {
                const int m = csc_pair[slot], i = csc_query[slot];
                const int r0 = clampr(rel[m * 3 + 0], L), r1 = clampr(rel[m * 3 + 1], L), r2 = clampr(rel[m * 3 + 2], L);
                float dl[HG], at[HG];
                float4 q4[HG], g4[HG];
#pragma unroll
                for (int t = 0; t < HG; t++) {  // no per-head guards: all loads of the pass together
                    const int te = min(t, hgn - 1);
                    dl[t] = grad_logit[(size_t)m * h + h0 + te];
                    at[t] = attn[(size_t)m * h + h0 + te];
                    q4[t] = ldg4(q + (size_t)i * C + (h0 + te) * D + 4 * c);
                    g4[t] = ldg4(go + (size_t)i * C + (h0 + te) * D + 4 * c);
                }
#pragma unroll
                for (int t = 0; t < HG; t++) {
                    a1[t] = fma4(dl[t], q4[t], a1[t]);
                    a2[t] = fma4(dl[t], tsum<D>(Tk, L, min(t, hgn - 1), r0, r1, r2, c), a2[t]);
                    av[t] = fma4(at[t], g4[t], av[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < HG; t++) {
            if (t < hgn) {
                const float4 t1 = xor_sum4<LPG, 64>(a1[t]), t2 = xor_sum4<LPG, 64>(a2[t]), tv = xor_sum4<LPG, 64>(av[t]);
                if (p == 0) {
                    stg4(grad_k + (size_t)kj * C + (h0 + t) * D + 4 * c, add4(t1, t2));
                    stg4(grad_v + (size_t)kj * C + (h0 + t) * D + 4 * c, tv);
                }
            }
        }
    }
}

// ---- launchers -------------------------------------------------------------------------------------------
static int walk_blocks(int rows, int groups, int waves_per_block, int per_cu) {
    int want = div_up(rows, waves_per_block);
    const int cus = usable_cus();
    int cap = cus * per_cu;
    if (groups > 1) cap = max(cus * per_cu / groups, cus / 2);
    return max(1, min(want, cap));
}

template <typename F>
static void with_heads(int h, F f) {
    if (h >= 3) f(std::integral_constant<int, 3>{});
    else if (h == 2) f(std::integral_constant<int, 2>{});
    else f(std::integral_constant<int, 1>{});
}

template <int TA>
static void launch_table_grad(int N, int h, int L, const float *w, const float *X, const int *offs, const int *pair_map,
                              const int *rel, float *grad_table, hipStream_t st) {
    with_heads(h, [&](auto tag) {
        constexpr int HG = decltype(tag)::value;
        const int groups = div_up(h, HG);
        using G = TableGeo<HG, TA>;
        const size_t lds = G::lds_bytes();
        if (getenv("P2_TG_STAMPS")) {  // diagnostic only: synchronous, prints the phase cycles of workgroup 0 to stderr
            unsigned long long *dbg = nullptr, host[TG_WAVES * 8];
            (void)hipMalloc(&dbg, sizeof(host));
            (void)hipMemset(dbg, 0, sizeof(host));
            const int bx = walk_blocks(N, groups, TG_WAVES, 2);
            hipLaunchKernelGGL((table_grad_kernel<HG, TA, true>), dim3(bx, groups), dim3(TG_WAVES * 64), lds, st, N, h, L, w, X, offs,
                               pair_map, rel, grad_table, rows_in_order(N), dbg);
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
            (void)hipFree(dbg);
            fprintf(stderr, "[tg stamps] N %d h %d blocks %d x %d %s: prologue ids relw atomics wait1 mfma wait2 flush (cycles)\n", N, h, bx, groups,
                    pair_map ? "by key" : "by query");
            for (int wv = 0; wv < TG_WAVES; wv += 5)
                fprintf(stderr, "[tg stamps]   wave %2d: %llu %llu %llu %llu %llu %llu %llu %llu\n", wv, host[wv * 8], host[wv * 8 + 1],
                        host[wv * 8 + 2], host[wv * 8 + 3], host[wv * 8 + 4], host[wv * 8 + 5], host[wv * 8 + 6], host[wv * 8 + 7]);
            return;
        }
        hipLaunchKernelGGL((table_grad_kernel<HG, TA>), dim3(walk_blocks(N, groups, TG_WAVES, 2), groups), dim3(TG_WAVES * 64), lds, st,
                           N, h, L, w, X, offs, pair_map, rel, grad_table, rows_in_order(N));
    });
}

// entry points used by rpe.hip; return false when the shape is outside this file's fast path
bool a2_bwd_mfma(int N, int NK, int M, int h, int hdim, int L, const float *go, const float *q, const int *offs, const float *k,
                 const float *table_q, const float *table_k, const int *rel, const int *co, const int *cp,
                 float *grad_q, float *grad_k, float *gtq, float *gtk) {
    if (hdim != 16 || co == nullptr || L < 1 || L > 80) return false;
    ForkJoin fj(state().stream, fork_worthwhile((int64_t)M * h));  // grad_q, grad_k and the two table gradients are independent
    with_heads(h, [&](auto tag) {
        constexpr int HG = decltype(tag)::value;
        const int groups = div_up(h, HG);
        const size_t lds = (size_t)HG * 3 * L * 16 * sizeof(float);
        hipLaunchKernelGGL((rows_table_sum_kernel<HG, true>), dim3(walk_blocks(NK, groups, 8, 4), groups), dim3(512), lds, fj.lane(0),
                           NK, h, L, go, co, cp, table_k, rel, grad_k, rows_in_order(NK));
        hipLaunchKernelGGL((rows_table_sum_kernel<HG, false>), dim3(walk_blocks(N, groups, 8, 4), groups), dim3(512), lds, fj.lane(1),
                           N, h, L, go, offs, (const int *)nullptr, table_q, rel, grad_q, rows_in_order(N));
    });
    if (L <= 64) {
        launch_table_grad<4>(NK, h, L, go, k, co, cp, rel, gtk, fj.lane(2));
        launch_table_grad<4>(N, h, L, go, q, offs, nullptr, rel, gtq, fj.lane(3));
    } else {
        launch_table_grad<5>(NK, h, L, go, k, co, cp, rel, gtk, fj.lane(2));
        launch_table_grad<5>(N, h, L, go, q, offs, nullptr, rel, gtq, fj.lane(3));
    }
    return true;
}

bool a4_bwd_mfma(int N, int h, int hdim, int L, const float *go, const int *offs, const int *idx1, const float *attn,
                 const float *v, const float *table, const int *rel, float *grad_attn, float *grad_table, ForkJoin &fj) {
    if (hdim != 16 || L < 1 || L > 80) return false;
    with_heads(h, [&](auto tag) {
        constexpr int HG = decltype(tag)::value;
        const int groups = div_up(h, HG);
        const size_t lds = (size_t)HG * 3 * L * 16 * sizeof(float);
        hipLaunchKernelGGL((a4_bwd_attn_kernel<HG>), dim3(walk_blocks(N, groups, 8, 4), groups), dim3(512), lds, fj.lane(0),
                           N, h, L, go, offs, idx1, v, table, rel, grad_attn, rows_in_order(N));
    });
    if (L <= 64) launch_table_grad<4>(N, h, L, attn, go, offs, nullptr, rel, grad_table, fj.lane(1));
    else launch_table_grad<5>(N, h, L, attn, go, offs, nullptr, rel, grad_table, fj.lane(1));
    return true;
}

// the optional fused backward; every output is fully written except the three table gradients (accumulated)
bool wattn_bwd(int N, int NK, int M, int h, int hdim, int L, const float *go, const float *q, const float *k, const float *v,
               const float *attn, const int *offs, const int *idx1, const float *table_q, const float *table_k,
               const float *table_v, const int *rel, const int *co, const int *cp, const int *cq, float *grad_logit,
               float *grad_q, float *grad_k, float *grad_v, float *gtq, float *gtk, float *gtv) {
    if (hdim != 16 || co == nullptr || L < 1 || L > 80) return false;
    hipStream_t st = state().stream;
    with_heads(h, [&](auto tag) {
        constexpr int HG = decltype(tag)::value;
        const int groups = div_up(h, HG);
        const size_t lds2 = (size_t)2 * HG * 3 * L * 16 * sizeof(float);
        allow_big_lds(wattn_bwd_query_kernel<HG>, lds2);
        hipLaunchKernelGGL((wattn_bwd_query_kernel<HG>), dim3(walk_blocks(N, groups, 8, 2), groups), dim3(512), lds2, st,
                           N, h, L, go, offs, idx1, attn, v, k, table_v, table_q, rel, grad_logit, grad_q, rows_in_order(N));
    });
    // everything below only reads grad_logit: the key walk and the three table gradients are independent
    ForkJoin fj(st, fork_worthwhile((int64_t)M * h));
    with_heads(h, [&](auto tag) {
        constexpr int HG = decltype(tag)::value;
        const int groups = div_up(h, HG);
        const size_t lds1 = (size_t)HG * 3 * L * 16 * sizeof(float);
        hipLaunchKernelGGL((wattn_bwd_key_kernel<HG>), dim3(walk_blocks(NK, groups, 8, 2), groups), dim3(512), lds1, fj.lane(0),
                           NK, h, L, grad_logit, attn, go, q, co, cp, cq, table_k, rel, grad_k, grad_v, rows_in_order(NK));
    });
    if (L <= 64) {
        launch_table_grad<4>(NK, h, L, grad_logit, k, co, cp, rel, gtk, fj.lane(1));
        launch_table_grad<4>(N, h, L, grad_logit, q, offs, nullptr, rel, gtq, fj.lane(2));
        launch_table_grad<4>(N, h, L, attn, go, offs, nullptr, rel, gtv, fj.lane(3));
    } else {
        launch_table_grad<5>(NK, h, L, grad_logit, k, co, cp, rel, gtk, fj.lane(1));
        launch_table_grad<5>(N, h, L, grad_logit, q, offs, nullptr, rel, gtq, fj.lane(2));
        launch_table_grad<5>(N, h, L, attn, go, offs, nullptr, rel, gtv, fj.lane(3));
    }
    return true;
}

}  // namespace p2
