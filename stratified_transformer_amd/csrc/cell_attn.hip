// Window-centric ("cell") attention, forward and backward, gfx950.
//
// The reference's three operators and every pair walker of rpe.hip / attention.hip gather a key row once per
// (query, key) pair although ~70 % of a pair list are dense window x window blocks
// (model/stratified_transformer.py:15-18) and the rest are shared by all queries of a (small window, large
// window) intersection (:20-38).  index.hip's cell plan states that structure: a CELL is a set of n_q queries
// that share one list of n_k candidate keys, i.e. a dense n_q x n_k tile of pairs, with a packed rel-pos index
// and a "not a key of this query" flag per tile entry.
//
// One wave per (cell, head).  Lane (p, c) = key slot p (16 per pass) x quarter c of the 16 floats of a head row,
// as in the pair walkers - but here a lane OWNS its keys for the whole cell: the key / value rows of up to
// 16 * CA_NP keys are loaded ONCE into registers (float4 per key slot) and reused by all n_q queries, and in the
// backward the key-side gradients dK / dV are register accumulators that leave the wave once per cell.  What
// remains per pair is what cannot be shared: the 9 table rows T(rel) from the LDS image of the head's three
// tables, the packed rel-pos word and the softmax weight.
//
// forward   sweep 1 (K in registers): logits -> softmax per query -> p stored to the cell-ordered buffer pbuf
//           sweep 2 (V in registers): out = sum p (v + Tv)
// backward  sweep A (V, dV in registers): grad_attn, softmax backward -> gs stored; dV
//           sweep B (K, dK in registers): dQ, dK
//           cell_table_grad_kernel: the three table gradients from p / gs (fixed-point LDS histograms per row,
//           outer products on the matrix cores: the scheme of rpe_bwd_mfma.hip on cell rows - a query's row of
//           the tile is contiguous, a key's column is strided by n_k - no pair map, no CSC)
// Cells with more than 16 * CA_NP keys are taken in chunks (a running max / sum per query in `ml`, logits parked
// in pbuf); the shipped configs never need more than one chunk.
#include "cell_common.h"

namespace p2 {

#ifdef CA_TRACE
__device__ unsigned long long ca_trace[3 * 1024];
#endif

// ---- storage type of q / k / v / tables: fp32, or bf16 (BASELINE config 3's second leg: bf16 storage, fp32 arithmetic;
// the reference's operators are fp32-only, stratified_transformer.py:183,194,208 `.float()`) ----
typedef unsigned short bf16_t;  // raw bits
__device__ __forceinline__ float4 ld_row4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 ld_row4(const bf16_t *p) {
    const uint2 u = *reinterpret_cast<const uint2 *>(p);  // four bf16: widening to fp32 is a shift
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float ld_elem(const float *p) { return *p; }
__device__ __forceinline__ float ld_elem(const bf16_t *p) { return __uint_as_float((unsigned)*p << 16); }
// image of one head's table: [axis][row][16] elements of T (global layout [L, h, 16, 3])
template <typename T>
__device__ __forceinline__ void stage_table_t(T *lds, const T *__restrict__ tab, int L, int h, int head) {
    const int total = 3 * L * 16;
    for (int x = threadIdx.x; x < total; x += blockDim.x) {
        const int i = x % 16, r = (x / 16) % L, ax = x / (16 * L);
        lds[x] = tab[(((size_t)r * h + head) * 16 + i) * 3 + ax];
    }
}

// ---- LDS image of a head's three tables: table TB at float offset TB * TS (TS a compile-time constant, so that the
// three tables of one (axis, row) differ by an immediate offset), inside a table [axis][row][16] ----
template <int LCAP>
struct TabGeo {
    static constexpr int TS = 3 * LCAP * 16;  // elements per table image
    static constexpr size_t bytes(size_t elem) { return (size_t)3 * TS * elem; }
};
struct RowOff {
    int o0, o1, o2;
};
__device__ __forceinline__ RowOff row_off(unsigned w, int L, int c) {
    RowOff r;
    r.o0 = (int)(w & 255u) * 16 + 4 * c;
    r.o1 = (int)(L + ((w >> 8) & 255u)) * 16 + 4 * c;
    r.o2 = (int)(2 * L + ((w >> 16) & 255u)) * 16 + 4 * c;
    return r;
}
// T(m)[4c..4c+3] = tab[r0,.,.,0] + tab[r1,.,.,1] + tab[r2,.,.,2]   (left to right, as the reference)
template <int OFF>
__device__ __forceinline__ float4 tsum_at(const float *lds, RowOff r) {
    return add4(add4(*reinterpret_cast<const float4 *>(lds + OFF + r.o0), *reinterpret_cast<const float4 *>(lds + OFF + r.o1)),
                *reinterpret_cast<const float4 *>(lds + OFF + r.o2));
}

// the three rows (one per axis) of one table for one pair, requested together
struct Rows3 {
    float4 a, b, c;
};
template <int OFF, typename T>
__device__ __forceinline__ Rows3 rows_at(const T *lds, RowOff r) {
    Rows3 x;
    x.a = ld_row4(lds + OFF + r.o0);
    x.b = ld_row4(lds + OFF + r.o1);
    x.c = ld_row4(lds + OFF + r.o2);
    return x;
}
// T(m)[4c..4c+3] = tab[r0,.,.,0] + tab[r1,.,.,1] + tab[r2,.,.,2]   (left to right, as the reference), as packed
// two-float adds (v_pk_add_f32: two lanes' worth of fp32 adds per instruction slot)
typedef float f32x2c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 padd4(float4 a, float4 b) {
    const f32x2c lo = f32x2c{a.x, a.y} + f32x2c{b.x, b.y}, hi = f32x2c{a.z, a.w} + f32x2c{b.z, b.w};
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 pfma4(float s, float4 a, float4 acc) {
    const f32x2c ss = f32x2c{s, s};
    const f32x2c lo = __builtin_elementwise_fma(ss, f32x2c{a.x, a.y}, f32x2c{acc.x, acc.y});
    const f32x2c hi = __builtin_elementwise_fma(ss, f32x2c{a.z, a.w}, f32x2c{acc.z, acc.w});
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}
__device__ __forceinline__ float4 rsum(Rows3 r) { return padd4(padd4(r.a, r.b), r.c); }
// <a, b> over this lane's four floats with packed multiply-adds (the two halves are summed at the end)
__device__ __forceinline__ f32x2c pdot_acc(float4 a, float4 b, f32x2c acc) {
    acc = __builtin_elementwise_fma(f32x2c{a.x, a.y}, f32x2c{b.x, b.y}, acc);
    return __builtin_elementwise_fma(f32x2c{a.z, a.w}, f32x2c{b.z, b.w}, acc);
}
__device__ __forceinline__ float pdot4(float4 a, float4 b) {
    const f32x2c r = pdot_acc(a, b, f32x2c{0.f, 0.f});
    return r.x + r.y;
}

// A chunk of a cell has np = 1..NP passes of 16 keys (wave-uniform).  The sweeps are straight-line code for a fixed
// number of passes - the table rows of pass t+1 are requested before pass t is consumed, which a per-pass branch
// would forbid (each pass a basic block of its own, every LDS round trip exposed) - instantiated for 2, 3, 4, 6 and 8
// passes; a chunk runs the smallest instance that holds it (slots past its end are masked anyway).
template <int NPMAX, typename F>
__device__ __forceinline__ void dispatch_passes(int np, F f) {
    if (np <= 2) f(std::integral_constant<int, 2>{});
    else if (np == 3) f(std::integral_constant<int, 3>{});
    else if (np == 4 || NPMAX <= 4) f(std::integral_constant<int, 4>{});
    else if constexpr (NPMAX > 4) {
        if (np <= 6) f(std::integral_constant<int, 6>{});
        else f(std::integral_constant<int, 8>{});
    }
}

template <typename T>
struct LaneCtx {
    const T *lds;
    int L, lane, p, c, head, h, C, hoff;
};
struct CellBufs {
    rsrc_t rel, key, qid;
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Key slots: lane (p, c) of an NPA-pass instance owns the NPA CONSECUTIVE keys j0 + p * NPA + t, t < NPA, of the chunk, so
// that its rel-pos words, softmax weights and logit gradients of one query are NPA consecutive dwords = one wide load.
// Query ids: 64 at a time, one per lane; a query's id is then a v_readlane away (a scalar: the q row address is uniform).
template <int NPA, typename T>
__device__ __forceinline__ void load_key_rows(const LaneCtx<T> &x, const CellBufs &cb, const T *__restrict__ rows, int j0, float4 (&r4)[NPA]) {
    unsigned keys[NPA];
    bload_words<NPA>(cb.key, (j0 + x.p * NPA) * 4, keys);  // (past the end: key 0, never used)
#pragma unroll
    for (int t = 0; t < NPA; t++) r4[t] = ld_row4(rows + (size_t)keys[t] * x.C + x.hoff);
}

template <int NPA, int TS, typename T>
__device__ __forceinline__ void fwd_sweep_logits(const LaneCtx<T> &x, const CellTask &ct, const CellBufs &cb, rsrc_t rs_p,
                                                 const T *__restrict__ q, const T *__restrict__ k, float *__restrict__ ml,
                                                 int ch, bool single, int j0, int nkc) {
    const int p = x.p, c = x.c;
    const int nvalid = min(max(nkc - p * NPA, 0), NPA);  // this lane's slots inside the chunk
    float4 k4[NPA];
    load_key_rows<NPA, T>(x, cb, k, j0, k4);
    // the inputs of query il+1 are requested while query il is worked on
    int ids = (int)bload_u32(cb.qid, x.lane * 4);
    int i_nx = __builtin_amdgcn_readlane(ids, 0);
    float4 q4_nx = ld_row4(q + (size_t)i_nx * x.C + x.hoff);
    unsigned w_nx[NPA];
    bload_words<NPA>(cb.rel, (j0 + p * NPA) * 4, w_nx);
    for (int il = 0; il < ct.nq; il++) {
        const int i = i_nx;
        const float4 q4 = q4_nx;
        unsigned w[NPA];
#pragma unroll
        for (int t = 0; t < NPA; t++) w[t] = w_nx[t];
        const int roff = (il * ct.nk + j0 + p * NPA) * 4;
        if (((il + 1) & 63) == 0) ids = (int)bload_u32(cb.qid, (il + 1 + x.lane) * 4);
        i_nx = __builtin_amdgcn_readlane(ids, (il + 1) & 63);  // (past the end: query 0, never used)
        q4_nx = ld_row4(q + (size_t)i_nx * x.C + x.hoff);
        bload_words<NPA>(cb.rel, roff + ct.nk * 4, w_nx);
        float lg[NPA];
        float mx = -INFINITY;
        RowOff ro = row_off(w[0], x.L, c);
        Rows3 rq = rows_at<0, T>(x.lds, ro), rk = rows_at<TS, T>(x.lds, ro);
#pragma unroll
        for (int t = 0; t < NPA; t++) {
            Rows3 rq1 = rq, rk1 = rk;
            if (t + 1 < NPA) {
                ro = row_off(w[t + 1], x.L, c);
                rq1 = rows_at<0, T>(x.lds, ro);
                rk1 = rows_at<TS, T>(x.lds, ro);
            }
            const f32x2c d2 = pdot_acc(k4[t], rsum(rk), pdot_acc(q4, rsum(rq), pdot_acc(q4, k4[t], f32x2c{0.f, 0.f})));
            const float s = quad_sum(d2.x + d2.y);
            lg[t] = (t < nvalid && !(w[t] >> 31)) ? s : -INFINITY;
            mx = fmaxf(mx, lg[t]);
            rq = rq1;
            rk = rk1;
            __builtin_amdgcn_sched_barrier(0);  // at most two passes' table rows in flight (register budget)
        }
        mx = slots_max16(mx);
        if (single) {
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NPA; t++) {
                lg[t] = __expf(lg[t] - mx);  // exp(-inf) = 0: flagged entries and slots past the end
                sum += lg[t];
            }
            const float inv = 1.0f / slots_sum16(sum);
#pragma unroll
            for (int t = 0; t < NPA; t++) lg[t] *= inv;
            if (c == 0) bstore_floats<NPA>(rs_p, roff, lg, nvalid);
        } else {
            float *st = ml + ((size_t)i * x.h + x.head) * 2;
            const float m_old = ch ? st[0] : -INFINITY, l_old = ch ? st[1] : 0.f;
            const float m_new = fmaxf(m_old, mx);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NPA; t++) sum += __expf(lg[t] - m_new);
            if (c == 0) bstore_floats<NPA>(rs_p, roff, lg, nvalid);  // logits; the second sweep makes them weights
            sum = slots_sum16(sum);
            if (x.lane == 0) {
                st[0] = m_new;
                st[1] = (m_old == -INFINITY ? 0.f : l_old * __expf(m_old - m_new)) + sum;
            }
        }
    }
}

template <int NPA, int TS, typename T>
__device__ __forceinline__ void fwd_sweep_values(const LaneCtx<T> &x, const CellTask &ct, const CellBufs &cb, rsrc_t rs_p,
                                                 const T *__restrict__ v, const float *__restrict__ ml, float *__restrict__ out,
                                                 int ch, bool single, int j0, int nkc) {
    const int p = x.p, c = x.c;
    const int nvalid = min(max(nkc - p * NPA, 0), NPA);
    float4 v4[NPA];
    load_key_rows<NPA, T>(x, cb, v, j0, v4);
    unsigned w_nx[NPA];
    float a_nx[NPA];
    bload_words<NPA>(cb.rel, (j0 + p * NPA) * 4, w_nx);
    bload_floats<NPA>(rs_p, (j0 + p * NPA) * 4, a_nx);
    int ids = (int)bload_u32(cb.qid, x.lane * 4);
    for (int il = 0; il < ct.nq; il++) {
        if (il && (il & 63) == 0) ids = (int)bload_u32(cb.qid, (il + x.lane) * 4);
        const int i = __builtin_amdgcn_readlane(ids, il & 63);
        const int roff = (il * ct.nk + j0 + p * NPA) * 4;
        unsigned w[NPA];
        float a[NPA];
#pragma unroll
        for (int t = 0; t < NPA; t++) {
            w[t] = w_nx[t];
            a[t] = a_nx[t];
        }
        bload_words<NPA>(cb.rel, roff + ct.nk * 4, w_nx);
        bload_floats<NPA>(rs_p, roff + ct.nk * 4, a_nx);
        if (!single) {
            const float *st = ml + ((size_t)i * x.h + x.head) * 2;
            const float m = st[0], inv = 1.0f / st[1];
#pragma unroll
            for (int t = 0; t < NPA; t++) a[t] = __expf(a[t] - m) * inv;
            if (c == 0) bstore_floats<NPA>(rs_p, roff, a, nvalid);
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        Rows3 rv = rows_at<2 * TS, T>(x.lds, row_off(w[0], x.L, c));
#pragma unroll
        for (int t = 0; t < NPA; t++) {
            Rows3 rv1 = rv;
            if (t + 1 < NPA) rv1 = rows_at<2 * TS, T>(x.lds, row_off(w[t + 1], x.L, c));
            const float at = t < nvalid ? a[t] : 0.f;  // (a flagged entry's weight is stored as 0)
            acc = pfma4(at, padd4(rsum(rv), v4[t]), acc);
            rv = rv1;
            __builtin_amdgcn_sched_barrier(0);
        }
        const float4 tot = slots_sum16_4(acc);
        if (p == 0) {
            float *o = out + (size_t)i * x.C + x.hoff;
            stg4(o, ch ? add4(tot, ldg4(o)) : tot);
        }
    }
}

template <int NP, int LCAP, typename T>
__global__ __launch_bounds__(CA_WAVES * 64) void cell_fwd_kernel(pointops2_cell_plan pl, int h, int L, const T *__restrict__ q,
                                                                 const T *__restrict__ k, const T *__restrict__ v,
                                                                 const T *__restrict__ table_q, const T *__restrict__ table_k,
                                                                 const T *__restrict__ table_v, float *__restrict__ out,
                                                                 float *__restrict__ ml, float *__restrict__ pbuf, size_t plane) {
    constexpr int D = 16, TS = TabGeo<LCAP>::TS;
    extern __shared__ float lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    LaneCtx<T> x;
    x.lds = lds;
    x.L = L;
    x.h = h;
    x.head = blockIdx.y;
    x.C = h * D;
    x.lane = threadIdx.x & 63;
    x.p = x.lane >> 2;
    x.c = x.lane & 3;
    x.hoff = x.head * D + 4 * x.c;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    stage_table_t<T>(lds, table_q, L, h, x.head);
    stage_table_t<T>(lds + TS, table_k, L, h, x.head);
    stage_table_t<T>(lds + 2 * TS, table_v, L, h, x.head);
    __syncthreads();
    const int nC = share_count(pl, pl.counts[0]);
    float *pb = pbuf + (size_t)x.head * plane;
#ifdef CA_TRACE  // diagnostic build (tools/interference.py): when and where every workgroup of the forward kernel ran
    const unsigned long long tr_t0 = wall_clock64();
#endif
    const int slots = gridDim.x * CA_WAVES, slot = blockIdx.x * CA_WAVES + wave;
    for (int round = 0; round * slots < nC; round++) {
        const int task = snake_task(round, slot, slots);
        if (task >= nC) continue;
        const CellTask ct = cell_task(pl, share_task(pl, task));
        const int nch = (ct.nk + 16 * NP - 1) / (16 * NP);
        const unsigned tile_bytes = (unsigned)ct.nq * ct.nk * 4u;
        CellBufs cb;
        cb.rel = make_rsrc(pl.relp + ct.pbase, tile_bytes);
        cb.key = make_rsrc(pl.cell_keys + ct.kb, (unsigned)ct.nk * 4u);
        cb.qid = make_rsrc(pl.cell_order + ct.qs, (unsigned)ct.nq * 4u);
        const rsrc_t rs_p = make_rsrc(pb + ct.pbase, tile_bytes);
        for (int ch = 0; ch < nch; ch++) {  // sweep 1: logits and softmax
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0);
            dispatch_passes<NP>((nkc + 15) >> 4, [&](auto tag) {
                fwd_sweep_logits<decltype(tag)::value, TS, T>(x, ct, cb, rs_p, q, k, ml, ch, nch == 1, j0, nkc);
            });
        }
#ifndef CA_SKIP_SWEEP2
        for (int ch = 0; ch < nch; ch++) {  // sweep 2: out = sum p (v + Tv)
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0);
            dispatch_passes<NP>((nkc + 15) >> 4, [&](auto tag) {
                fwd_sweep_values<decltype(tag)::value, TS, T>(x, ct, cb, rs_p, v, ml, out, ch, nch == 1, j0, nkc);
            });
        }
#endif
    }
#ifdef CA_TRACE
    __syncthreads();
    if (threadIdx.x == 0) {
        const int wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (wg < 1024) {
            ca_trace[wg * 3 + 0] = tr_t0;
            ca_trace[wg * 3 + 1] = wall_clock64();
            ca_trace[wg * 3 + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);  // HW_ID, XCC_ID
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// backward, sweeps A and B
// ------------------------------------------------------------------------------------------------
// adds the key-side accumulators of pass t (lane (p, c): floats 4c..4c+3 of the key j0 + p * NPA + t) to grad[key, head, :]:
// through a wave-private LDS tile, so that one atomic instruction covers four whole 64-byte head rows
template <int NPA, typename T>
__device__ __forceinline__ void flush_key_pass(float *scr, float4 acc, rsrc_t rs_key, int j0, int t, int nkc, float *__restrict__ grad,
                                               const LaneCtx<T> &x) {
    *reinterpret_cast<float4 *>(scr + x.p * 16 + 4 * x.c) = acc;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int s = (x.lane >> 4) + 4 * kk;
        const int jl = s * NPA + t;
        const int key = (int)bload_u32(rs_key, (j0 + jl) * 4);
#ifndef CA_SKIP_FLUSH  // (experiment: what the dK / dV atomics cost)
        if (jl < nkc) unsafeAtomicAdd(grad + (size_t)key * x.C + x.head * 16 + (x.lane & 15), scr[s * 16 + (x.lane & 15)]);
#endif
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
}

// sweep A: grad_attn = <go, v + Tv>, gs = p (grad_attn - <go, out>) stored, dV += p go
template <int NPA, int TS, typename T>
__device__ __forceinline__ void bwd_sweep_values(const LaneCtx<T> &x, const CellTask &ct, const CellBufs &cb, rsrc_t rs_p, rsrc_t rs_g,
                                                 float *scr, const float *__restrict__ go, const float *__restrict__ out,
                                                 const T *__restrict__ v, float *__restrict__ grad_v, int j0, int nkc) {
    const int p = x.p, c = x.c;
    const int nvalid = min(max(nkc - p * NPA, 0), NPA);
    float4 v4[NPA], dv4[NPA];
    load_key_rows<NPA, T>(x, cb, v, j0, v4);
#pragma unroll
    for (int t = 0; t < NPA; t++) dv4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned w_nx[NPA];
    float a_nx[NPA];
    bload_words<NPA>(cb.rel, (j0 + p * NPA) * 4, w_nx);
    bload_floats<NPA>(rs_p, (j0 + p * NPA) * 4, a_nx);
    int ids = (int)bload_u32(cb.qid, x.lane * 4);
    int i_nx = __builtin_amdgcn_readlane(ids, 0);
    float4 go_nx = ldg4(go + (size_t)i_nx * x.C + x.hoff), o_nx = ldg4(out + (size_t)i_nx * x.C + x.hoff);
    for (int il = 0; il < ct.nq; il++) {
        const int roff = (il * ct.nk + j0 + p * NPA) * 4;
        const float4 go4 = go_nx, o4 = o_nx;
        unsigned w[NPA];
        float a[NPA];
#pragma unroll
        for (int t = 0; t < NPA; t++) {
            w[t] = w_nx[t];
            a[t] = t < nvalid ? a_nx[t] : 0.f;
        }
        if (((il + 1) & 63) == 0) ids = (int)bload_u32(cb.qid, (il + 1 + x.lane) * 4);
        i_nx = __builtin_amdgcn_readlane(ids, (il + 1) & 63);
        go_nx = ldg4(go + (size_t)i_nx * x.C + x.hoff);
        o_nx = ldg4(out + (size_t)i_nx * x.C + x.hoff);
        bload_words<NPA>(cb.rel, roff + ct.nk * 4, w_nx);
        bload_floats<NPA>(rs_p, roff + ct.nk * 4, a_nx);
        const float delta = quad_sum(pdot4(go4, o4));  // = sum over the row of p * grad_attn
        float gs[NPA];
        Rows3 rv = rows_at<2 * TS, T>(x.lds, row_off(w[0], x.L, c));
#pragma unroll
        for (int t = 0; t < NPA; t++) {
            Rows3 rv1 = rv;
            if (t + 1 < NPA) rv1 = rows_at<2 * TS, T>(x.lds, row_off(w[t + 1], x.L, c));
            const float ga = quad_sum(pdot4(go4, padd4(rsum(rv), v4[t])));
            gs[t] = a[t] * (ga - delta);
            dv4[t] = pfma4(a[t], go4, dv4[t]);
            rv = rv1;
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c == 0) bstore_floats<NPA>(rs_g, roff, gs, nvalid);
    }
#pragma unroll
    for (int t = 0; t < NPA; t++) flush_key_pass<NPA, T>(scr, dv4[t], cb.key, j0, t, nkc, grad_v, x);
}

// sweep B: dQ = sum gs (k + Tq), dK += gs (q + Tk)
template <int NPA, int TS, typename T>
__device__ __forceinline__ void bwd_sweep_keys(const LaneCtx<T> &x, const CellTask &ct, const CellBufs &cb, rsrc_t rs_g, float *scr,
                                               const T *__restrict__ q, const T *__restrict__ k, float *__restrict__ grad_q,
                                               float *__restrict__ grad_k, int ch, int j0, int nkc) {
    const int p = x.p, c = x.c;
    const int nvalid = min(max(nkc - p * NPA, 0), NPA);
    float4 k4[NPA], dk4[NPA];
    load_key_rows<NPA, T>(x, cb, k, j0, k4);
#pragma unroll
    for (int t = 0; t < NPA; t++) dk4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned w_nx[NPA];
    float g_nx[NPA];
    bload_words<NPA>(cb.rel, (j0 + p * NPA) * 4, w_nx);
    bload_floats<NPA>(rs_g, (j0 + p * NPA) * 4, g_nx);
    int ids = (int)bload_u32(cb.qid, x.lane * 4);
    int i_nx = __builtin_amdgcn_readlane(ids, 0);
    float4 q_nx = ld_row4(q + (size_t)i_nx * x.C + x.hoff);
    for (int il = 0; il < ct.nq; il++) {
        const int i = i_nx;
        const int roff = (il * ct.nk + j0 + p * NPA) * 4;
        const float4 q4 = q_nx;
        unsigned w[NPA];
        float g[NPA];
#pragma unroll
        for (int t = 0; t < NPA; t++) {
            w[t] = w_nx[t];
            g[t] = t < nvalid ? g_nx[t] : 0.f;
        }
        if (((il + 1) & 63) == 0) ids = (int)bload_u32(cb.qid, (il + 1 + x.lane) * 4);
        i_nx = __builtin_amdgcn_readlane(ids, (il + 1) & 63);
        q_nx = ld_row4(q + (size_t)i_nx * x.C + x.hoff);
        bload_words<NPA>(cb.rel, roff + ct.nk * 4, w_nx);
        bload_floats<NPA>(rs_g, roff + ct.nk * 4, g_nx);
        float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
        RowOff ro = row_off(w[0], x.L, c);
        Rows3 rq = rows_at<0, T>(x.lds, ro), rk = rows_at<TS, T>(x.lds, ro);
#pragma unroll
        for (int t = 0; t < NPA; t++) {
            Rows3 rq1 = rq, rk1 = rk;
            if (t + 1 < NPA) {
                ro = row_off(w[t + 1], x.L, c);
                rq1 = rows_at<0, T>(x.lds, ro);
                rk1 = rows_at<TS, T>(x.lds, ro);
            }
            dq = pfma4(g[t], padd4(rsum(rq), k4[t]), dq);
            dk4[t] = pfma4(g[t], padd4(rsum(rk), q4), dk4[t]);
            rq = rq1;
            rk = rk1;
            __builtin_amdgcn_sched_barrier(0);
        }
        const float4 tot = slots_sum16_4(dq);
        if (p == 0) {
            float *o = grad_q + (size_t)i * x.C + x.hoff;
            stg4(o, ch ? add4(tot, ldg4(o)) : tot);
        }
    }
#pragma unroll
    for (int t = 0; t < NPA; t++) flush_key_pass<NPA, T>(scr, dk4[t], cb.key, j0, t, nkc, grad_k, x);
}

template <int NP, int LCAP, typename T>
__global__ __launch_bounds__(CA_WAVES_BWD * 64) void cell_bwd_kernel(pointops2_cell_plan pl, int h, int L, const float *__restrict__ go,
                                                                 const T *__restrict__ q, const T *__restrict__ k,
                                                                 const T *__restrict__ v, const float *__restrict__ out,
                                                                 const T *__restrict__ table_q, const T *__restrict__ table_k,
                                                                 const T *__restrict__ table_v, const float *__restrict__ pbuf,
                                                                 float *__restrict__ gsbuf, size_t plane, float *__restrict__ grad_q,
                                                                 float *__restrict__ grad_k, float *__restrict__ grad_v) {
    constexpr int D = 16, TS = TabGeo<LCAP>::TS;
    extern __shared__ float lds_raw[];
    T *lds = reinterpret_cast<T *>(lds_raw);
    LaneCtx<T> x;
    x.lds = lds;
    x.L = L;
    x.h = h;
    x.head = blockIdx.y;
    x.C = h * D;
    x.lane = threadIdx.x & 63;
    x.p = x.lane >> 2;
    x.c = x.lane & 3;
    x.hoff = x.head * D + 4 * x.c;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *scr = reinterpret_cast<float *>(lds + 3 * TS) + wave * 256;
    stage_table_t<T>(lds, table_q, L, h, x.head);
    stage_table_t<T>(lds + TS, table_k, L, h, x.head);
    stage_table_t<T>(lds + 2 * TS, table_v, L, h, x.head);
    __syncthreads();
    const int nC = share_count(pl, pl.counts[0]);
    const float *pb = pbuf + (size_t)x.head * plane;
    float *gb = gsbuf + (size_t)x.head * plane;
    const int slots = gridDim.x * CA_WAVES_BWD, slot = blockIdx.x * CA_WAVES_BWD + wave;
    for (int round = 0; round * slots < nC; round++) {
        const int task = snake_task(round, slot, slots);
        if (task >= nC) continue;
        const CellTask ct = cell_task(pl, share_task(pl, task));
        const int nch = (ct.nk + 16 * NP - 1) / (16 * NP);
        const unsigned tile_bytes = (unsigned)ct.nq * ct.nk * 4u;
        CellBufs cb;
        cb.rel = make_rsrc(pl.relp + ct.pbase, tile_bytes);
        cb.key = make_rsrc(pl.cell_keys + ct.kb, (unsigned)ct.nk * 4u);
        cb.qid = make_rsrc(pl.cell_order + ct.qs, (unsigned)ct.nq * 4u);
        const rsrc_t rs_p = make_rsrc(pb + ct.pbase, tile_bytes);
        const rsrc_t rs_g = make_rsrc(gb + ct.pbase, tile_bytes);
        for (int ch = 0; ch < nch; ch++) {
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0);
            dispatch_passes<NP>((nkc + 15) >> 4, [&](auto tag) {
                bwd_sweep_values<decltype(tag)::value, TS, T>(x, ct, cb, rs_p, rs_g, scr, go, out, v, grad_v, j0, nkc);
            });
        }
        for (int ch = 0; ch < nch; ch++) {
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0);
            dispatch_passes<NP>((nkc + 15) >> 4, [&](auto tag) {
                bwd_sweep_keys<decltype(tag)::value, TS, T>(x, ct, cb, rs_g, scr, q, k, grad_q, grad_k, ch, j0, nkc);
            });
        }
    }
}

// ------------------------------------------------------------------------------------------------
// table gradients on cell tiles.  grad_table[r, head, i, ax] += sum over tile entries with rel[ax] == r of w * X[row point, head, i]
//   BYKEY = false: a row = a query (its row of the tile), X = q (w = logit gradients) or grad_out (w = softmax weights)
//   BYKEY = true:  a row = a key   (its column of the tile), X = k (w = logit gradients)
// The sum is factored per row as in rpe_bwd_mfma.hip: a 3 x L histogram of the row's weights by rel index (32-bit fixed
// point: LDS integer atomics run at LDS write speed, float ones at ~1 lane per 3 cycles), times the row's X vector - an
// outer product, taken on the matrix cores four rows at a time: D[bin, i] += A[bin, row] * B[row, i] is one
// v_mfma_f32_16x16x4_f32 per 16-bin tile and axis.  Here a wave works alone: it takes FOUR ADJACENT ROWS of a cell at
// once (lane (p, c): entry p of a pass, row c of the four - for keys the four columns are 16 contiguous bytes per query),
// keeps the whole table slice of its head in accumulators (TA x 3 tiles) and meets the other waves of its workgroup only
// at the end, when the accumulators are summed through LDS and leave as contiguous atomics.  No row claims, no
// barriers, no descriptor chains in the loop: a cell's rows share one set of scalars.
// ------------------------------------------------------------------------------------------------
struct CFixScale {
    float mul, inv;
};
__device__ __forceinline__ CFixScale c_row_scale(unsigned maxbits, int n) {
    CFixScale sc;
    if (maxbits >= 0x7f800000u) {  // Inf / NaN among the weights
        sc.mul = 0.f;
        sc.inv = __uint_as_float(0x7fc00000u);
        return sc;
    }
    const int E = (int)(maxbits >> 23) - 126;  // max|w| < 2^E
    const int bits_n = 32 - __clz(max(n, 1));  // n < 2^bits_n
    const int S = max(-126, min(126, 30 - bits_n - E));
    sc.mul = __uint_as_float((unsigned)(S + 127) << 23);
    sc.inv = __uint_as_float((unsigned)(127 - S) << 23);
    return sc;
}

#ifndef CT_WAVES_OVERRIDE
#define CT_WAVES_OVERRIDE 8  // (12 in round 2; with the grid below 8 measured 5 % less backward time per step: tools/bench_cell.py, round 3)
#endif
constexpr int CT_WAVES = CT_WAVES_OVERRIDE;

template <int TA>
struct CellTableGeo {
    static constexpr int LP = TA * 16;       // padded bins per axis
    static constexpr int ROW = 3 * LP + 16;  // ints per histogram row (+16: the four rows of one read on different banks)
    static constexpr size_t walk_bytes() { return (size_t)CT_WAVES * 4 * ROW * 4; }
    static constexpr size_t sum_bytes() { return (size_t)LP * 48 * 4; }
    static constexpr size_t lds_bytes() { return walk_bytes() > sum_bytes() ? walk_bytes() : sum_bytes(); }
};

template <int TA, bool BYKEY, typename XT>
__device__ __forceinline__ void cell_table_grad_body(const pointops2_cell_plan &pl, int h, int L, const float *__restrict__ wbuf, size_t plane,
                                                     const XT *__restrict__ X, float *__restrict__ grad_table) {
    constexpr int D = 16;
    constexpr int MAXP = BYKEY ? 4 : 8;  // passes of 16 entries per row a segment holds in registers (a key's column is short)
    using G = CellTableGeo<TA>;
    extern __shared__ float lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int C = h * D, head = blockIdx.y;
    const int p = lane >> 2, c = lane & 3;     // entry of a pass, row of the four
    const int kq = lane >> 4, col = lane & 15;  // MFMA operand coordinates: row of the four, bin / feature
    int *hist = reinterpret_cast<int *>(lds) + wave * 4 * G::ROW;  // [4][ROW], private to the wave
    const float *wb = wbuf + (size_t)head * plane;

    f32x4c acc[TA][3];
#pragma unroll
    for (int bt = 0; bt < TA; bt++)
#pragma unroll
        for (int ax = 0; ax < 3; ax++) acc[bt][ax] = f32x4c{0.f, 0.f, 0.f, 0.f};
    for (int x = lane; x < 4 * G::ROW; x += 64) hist[x] = 0;

    // rows = queries: a task is a cell (piece), largest first; rows = keys: a task is a parent (all pieces of an uncut cell:
    // one [sum n_q, n_k] tile), so that a key's column is as long as the cut allows
    // (a share of the cells - one scene over several ranks - applies to the query side; the key side walks every parent, and the
    //  caller zero-fills the weight planes so that the pieces of other ranks contribute nothing)
    const int nC = BYKEY ? pl.counts[4] : share_count(pl, pl.counts[0]);
    const int slots = gridDim.x * CT_WAVES, slot = blockIdx.x * CT_WAVES + wave;
    for (int round = 0; round * slots < nC; round++) {
        const int task = snake_task(round, slot, slots);
        if (task >= nC) continue;
        CellTask ct;
        if (BYKEY) {
            const int c0 = __builtin_amdgcn_readfirstlane(pl.parent_first[task]), c1 = __builtin_amdgcn_readfirstlane(pl.parent_first[task + 1]);
            ct.qs = __builtin_amdgcn_readfirstlane(pl.cell_qstart[c0]);
            ct.nq = __builtin_amdgcn_readfirstlane(pl.cell_qstart[c1]) - ct.qs;
            ct.kb = __builtin_amdgcn_readfirstlane(pl.cell_kbase[c0]);
            ct.nk = __builtin_amdgcn_readfirstlane(pl.cell_kbase[c0 + 1]) - ct.kb;
            ct.pbase = __builtin_amdgcn_readfirstlane(pl.cell_pbase[c0]);
        } else {
            ct = cell_task(pl, share_task(pl, task));
        }
        const unsigned tile_bytes = (unsigned)ct.nq * ct.nk * 4u;
        const rsrc_t rs_rel = make_rsrc(pl.relp + ct.pbase, tile_bytes);
        const rsrc_t rs_w = make_rsrc(wb + ct.pbase, tile_bytes);
        const rsrc_t rs_row = BYKEY ? make_rsrc(pl.cell_keys + ct.kb, (unsigned)ct.nk * 4u) : make_rsrc(pl.cell_order + ct.qs, (unsigned)ct.nq * 4u);
        const int nrows = BYKEY ? ct.nk : ct.nq, nent = BYKEY ? ct.nq : ct.nk;
        // entry e of row r0 + c: rows = keys: tile[e][r0 + c];  rows = queries: tile[r0 + c][e]
        auto entry_off = [&](int r0, int ent) -> int { return (BYKEY ? ent * ct.nk + r0 + c : (r0 + c) * ct.nk + ent) * 4; };
        // one segment (<= 16 * MAXP entries per row) of the four rows r0..r0+3: histograms, then the outer products
        auto consume = [&](int r0, int e0, float xv, unsigned (&wr)[MAXP], float (&wt)[MAXP]) {
            const int ne = min(nent - e0, 16 * MAXP);
            const bool row_ok = r0 + c < nrows;
            unsigned mxb = 0u;
#pragma unroll
            for (int t = 0; t < MAXP; t++) {
                const bool live = row_ok && e0 + 16 * t + p < nent && !(wr[t] >> 31);
                wt[t] = live ? wt[t] : 0.f;
                mxb = max(mxb, __float_as_uint(fabsf(wt[t])));
            }
            const CFixScale sc = c_row_scale(wave_max_u32(mxb), ne);
            int *myh = hist + c * G::ROW;
#pragma unroll
            for (int t = 0; t < MAXP; t++) {
                if (16 * t < ne && wt[t] != 0.f) {
                    const int v = __float2int_rn(wt[t] * sc.mul);
                    atomicAdd(&myh[wr[t] & 255u], v);
                    atomicAdd(&myh[G::LP + ((wr[t] >> 8) & 255u)], v);
                    atomicAdd(&myh[2 * G::LP + ((wr[t] >> 16) & 255u)], v);
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            const float b = xv * sc.inv;
#pragma unroll
            for (int bt = 0; bt < TA; bt++)
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    // read the bin and clear it for the next segment in one LDS operation
                    const float a = (float)__hip_atomic_exchange(&hist[kq * G::ROW + ax * G::LP + bt * 16 + col], 0, __ATOMIC_RELAXED,
                                                                 __HIP_MEMORY_SCOPE_WORKGROUP);
                    acc[bt][ax] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[bt][ax], 0, 0, 0);
                }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        };
        // Software pipeline over the cell's row quads: while quad q is consumed, the first segment and the X rows of quad
        // q+1 and the row ids of quad q+2 are in flight (an id, the X row it addresses and the tile entries would otherwise
        // be two dependent memory round trips per quad).  Reads past the end of a buffer return 0 and are never used.
        unsigned wr_n[MAXP];
        float wt_n[MAXP];
#pragma unroll
        for (int t = 0; t < MAXP; t++) {
            wr_n[t] = 0x80000000u;
            wt_n[t] = 0.f;
            if (16 * t < nent) {  // wave-uniform
                wr_n[t] = bload_u32(rs_rel, entry_off(0, 16 * t + p));
                wt_n[t] = bload_f32(rs_w, entry_off(0, 16 * t + p));
            }
        }
        int pt_n = (int)bload_u32(rs_row, kq * 4);
        float xv_n = ld_elem(X + (size_t)pt_n * C + head * D + col);  // B operand: X[point of row r0 + kq, head, col]
        pt_n = (int)bload_u32(rs_row, (4 + kq) * 4);
        for (int r0 = 0; r0 < nrows; r0 += 4) {
            unsigned wr[MAXP];
            float wt[MAXP];
#pragma unroll
            for (int t = 0; t < MAXP; t++) {
                wr[t] = wr_n[t];
                wt[t] = wt_n[t];
            }
            const float xv = r0 + kq < nrows ? xv_n : 0.f;
            xv_n = ld_elem(X + (size_t)pt_n * C + head * D + col);
            pt_n = (int)bload_u32(rs_row, (r0 + 8 + kq) * 4);
#pragma unroll
            for (int t = 0; t < MAXP; t++)
                if (16 * t < nent) {
                    wr_n[t] = bload_u32(rs_rel, entry_off(r0 + 4, 16 * t + p));
                    wt_n[t] = bload_f32(rs_w, entry_off(r0 + 4, 16 * t + p));
                }
            consume(r0, 0, xv, wr, wt);
            for (int e0 = 16 * MAXP; e0 < nent; e0 += 16 * MAXP) {  // rows longer than one segment (rare)
#pragma unroll
                for (int t = 0; t < MAXP; t++) {
                    wr[t] = bload_u32(rs_rel, entry_off(r0, e0 + 16 * t + p));
                    wt[t] = bload_f32(rs_w, entry_off(r0, e0 + 16 * t + p));
                }
                consume(r0, e0, xv, wr, wt);
            }
        }
    }
    // Sum over the workgroup's waves through one LDS image in table order ([bin][16][3]: the 48 floats of a (bin, head) are
    // contiguous in the [L, h, 16, 3] table), then contiguous atomics.  C/D layout of 16x16x4: lane holds
    // D[row = (lane >> 4) * 4 + reg][col = lane & 15]  (row = bin, col = feature).
    __syncthreads();  // every wave is done with its histograms
    float *sum = lds;
    for (int w = 0; w < CT_WAVES; w++) {
        if (wave == w) {
#pragma unroll
            for (int bt = 0; bt < TA; bt++)
#pragma unroll
                for (int ax = 0; ax < 3; ax++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) {
                        float *d = &sum[((bt * 16 + kq * 4 + reg) * 16 + col) * 3 + ax];
                        *d = w ? *d + acc[bt][ax][reg] : acc[bt][ax][reg];
                    }
        }
        __syncthreads();
    }
    for (int x = threadIdx.x; x < L * 48; x += CT_WAVES * 64) {
        const float val = sum[x];
        if (val != 0.f) unsafeAtomicAdd(grad_table + ((size_t)(x / 48) * h + head) * 48 + x % 48, val);
    }
}

template <int TA, bool BYKEY, typename XT>
__global__ __launch_bounds__(CT_WAVES * 64) void cell_table_grad_kernel(pointops2_cell_plan pl, int h, int L, const float *__restrict__ wbuf,
                                                                        size_t plane, const XT *__restrict__ X,
                                                                        float *__restrict__ grad_table) {
    cell_table_grad_body<TA, BYKEY, XT>(pl, h, L, wbuf, plane, X, grad_table);
}

// the three table gradients of a block as ONE grid (blockIdx.z: 0 = key side, the longest, first in dispatch order; 1 = query side;
// 2 = value side): each of them alone is short of independent work on the small stages, together they overlap
template <int TA, typename T>
__global__ __launch_bounds__(CT_WAVES * 64) void cell_table_grad3_kernel(pointops2_cell_plan pl, int h, int L, const float *__restrict__ gsbuf,
                                                                         const float *__restrict__ pbuf, size_t plane,
                                                                         const T *__restrict__ q, const T *__restrict__ k,
                                                                         const float *__restrict__ grad_out, float *__restrict__ gtq,
                                                                         float *__restrict__ gtk, float *__restrict__ gtv) {
    if (blockIdx.z == 0) cell_table_grad_body<TA, true, T>(pl, h, L, gsbuf, plane, k, gtk);
    else if (blockIdx.z == 1) cell_table_grad_body<TA, false, T>(pl, h, L, gsbuf, plane, q, gtq);
    else cell_table_grad_body<TA, false, float>(pl, h, L, pbuf, plane, grad_out, gtv);
}

static int device_cus() { return num_cus(); }
#ifdef CA_TRACE
} // namespace p2
extern "C" void pointops2_diag_read_cell_trace(unsigned long long *host) { (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(p2::ca_trace), sizeof(unsigned long long) * 3 * 1024); }
namespace p2 {
#endif

template <typename T>
static void launch_cell_fwd(const pointops2_cell_plan *plan, int h, int hdim, int L, const T *q, const T *k, const T *v, const T *table_q,
                            const T *table_k, const T *table_v, float *out, float *ml, float *pbuf) {
    if (plan == nullptr || plan->n_points <= 0) return;
    if (hdim != 16) { set_error("cell_attention: d != 16"); return; }
    if (L < 1) { set_error("cell_attention: no table rows"); return; }
    // relp's indices were clamped to [0, plan->table_rows) and L is the axis stride of the LDS table image
    if (L != plan->table_rows) { set_error("cell_attention: the tables' row count differs from the plan's table_rows"); return; }
    if constexpr (std::is_same<T, float>::value) {
        // the matrix-core forward (cell_attn_mfma.hip); P2_CELL_MFMA=0 keeps the VALU walkers below
        if (cell_fwd_mfma_launch(plan, h, L, q, k, v, table_q, table_k, table_v, out, pbuf)) {
            check_launch();
            return;
        }
    }
    const dim3 block(CA_WAVES * 64);
    const size_t plane = (size_t)plan->n_pairs;
    if (L <= 80) {
        const size_t lds = TabGeo<80>::bytes(sizeof(T));
        allow_big_lds(cell_fwd_kernel<CA_NP, 80, T>, lds);
        const dim3 grid(cell_grid_x(1, plan->n_cells, h, CA_WAVES), h);
        hipLaunchKernelGGL((cell_fwd_kernel<CA_NP, 80, T>), grid, block, lds, state().stream, *plan, h, L, q, k, v, table_q, table_k, table_v, out, ml,
                           pbuf, plane);
    } else if (L <= 160) {
        const size_t lds = TabGeo<160>::bytes(sizeof(T));
        allow_big_lds(cell_fwd_kernel<CA_NP, 160, T>, lds);
        const dim3 grid(cell_grid_x(1, plan->n_cells, h, CA_WAVES), h);
        hipLaunchKernelGGL((cell_fwd_kernel<CA_NP, 160, T>), grid, block, lds, state().stream, *plan, h, L, q, k, v, table_q, table_k, table_v, out, ml,
                           pbuf, plane);
    } else {
        set_error("cell_attention: more than 160 table rows (use the operators)");
        return;
    }
    check_launch();
}

template <typename T>
static void launch_cell_bwd(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *grad_out, const T *q, const T *k, const T *v,
                            const float *out, const T *table_q, const T *table_k, const T *table_v, const float *pbuf, float *gsbuf, float *grad_q,
                            float *grad_k, float *grad_v, float *grad_table_q, float *grad_table_k, float *grad_table_v) {
    if (plan == nullptr || plan->n_points <= 0) return;
    if (hdim != 16) { set_error("cell_attention: d != 16"); return; }
    if (L < 1 || L > 80) { set_error("cell_attention backward: table rows L must be in 1..80"); return; }
    if (L != plan->table_rows) { set_error("cell_attention backward: the tables' row count differs from the plan's table_rows"); return; }
    hipStream_t st = state().stream;
    const size_t lds = TabGeo<80>::bytes(sizeof(T)) + (size_t)CA_WAVES_BWD * 256 * sizeof(float);
    allow_big_lds(cell_bwd_kernel<CA_NP_BWD, 80, T>, lds);
    const size_t plane = (size_t)plan->n_pairs;
    static const int cb_div = getenv("P2_CB_DIV") ? atoi(getenv("P2_CB_DIV")) : 1;
    hipLaunchKernelGGL((cell_bwd_kernel<CA_NP_BWD, 80, T>), dim3(std::max(1, cell_grid_x(1, plan->n_cells, h, CA_WAVES_BWD) / cb_div), h),
                       dim3(CA_WAVES_BWD * 64), lds, st, *plan, h, L, grad_out, q, k, v, out, table_q, table_k, table_v, pbuf, gsbuf, plane, grad_q,
                       grad_k, grad_v);
    // the three table gradients read p / gs only
    // Grid of the table-gradient bodies: ONE workgroup of 8 waves per free CU and body (round 2: two of 12).
    // Measured (tools/bench_cell.py, backward of a block, us, two -> one -> half): stage 0 784 -> 717 -> 725 / 905 -> 831 -> 833 (plain / shifted
    // pattern), stage 1 512 -> 448 -> 433 / 505 -> 452 -> 423, stage 2 381 -> 345 -> 304 / 361 -> 302 -> 292, stage 3 317 -> 301 -> 303 /
    // 265 -> 223 -> 212: every workgroup pays a fixed zero-fill, a 12-round reduction through LDS and a 3 072-float atomic flush per body, and
    // the bodies are latency-bound - more resident workgroups only add to both.  (Halving the grids of cell_fwd / cell_bwd the same way
    // costs 40-70 %: those are bound by resident waves.)  P2_CT_PER_CU / P2_CT_DIV override.
    static const int ct_per_cu = getenv("P2_CT_PER_CU") ? atoi(getenv("P2_CT_PER_CU")) : 1;
    static const int ct_div_env = getenv("P2_CT_DIV") ? atoi(getenv("P2_CT_DIV")) : 0;
    const int ct_div = ct_div_env > 0 ? ct_div_env : 1;  // (with 12 waves per workgroup, halving the grid on the smaller stages paid; with 8 it does not)
    const int gx_t = std::max(1, cell_grid_x(ct_per_cu, plan->n_cells, h, CT_WAVES) / ct_div);
    const dim3 tgrid(gx_t, h), tblock(CT_WAVES * 64);
    // one grid for the three (P2_CELL_TABLE3=0: three launches in a row): backward of a block 10-120 us shorter, most on the small stages
    static const bool one_grid = getenv("P2_CELL_TABLE3") == nullptr || atoi(getenv("P2_CELL_TABLE3")) != 0;
    if (one_grid) {
        const dim3 grid3(gx_t, h, 3);
        if (L <= 64) hipLaunchKernelGGL((cell_table_grad3_kernel<4, T>), grid3, tblock, CellTableGeo<4>::lds_bytes(), st, *plan, h, L, gsbuf, pbuf, plane, q, k, grad_out,
                                        grad_table_q, grad_table_k, grad_table_v);
        else hipLaunchKernelGGL((cell_table_grad3_kernel<5, T>), grid3, tblock, CellTableGeo<5>::lds_bytes(), st, *plan, h, L, gsbuf, pbuf, plane, q, k, grad_out,
                                grad_table_q, grad_table_k, grad_table_v);
        check_launch();
        return;
    }
    if (L <= 64) {
        using G = CellTableGeo<4>;
        hipLaunchKernelGGL((cell_table_grad_kernel<4, false, T>), tgrid, tblock, G::lds_bytes(), st, *plan, h, L, gsbuf, plane, q, grad_table_q);
        hipLaunchKernelGGL((cell_table_grad_kernel<4, false, float>), tgrid, tblock, G::lds_bytes(), st, *plan, h, L, pbuf, plane, grad_out, grad_table_v);
        hipLaunchKernelGGL((cell_table_grad_kernel<4, true, T>), tgrid, tblock, G::lds_bytes(), st, *plan, h, L, gsbuf, plane, k, grad_table_k);
    } else {
        using G = CellTableGeo<5>;
        hipLaunchKernelGGL((cell_table_grad_kernel<5, false, T>), tgrid, tblock, G::lds_bytes(), st, *plan, h, L, gsbuf, plane, q, grad_table_q);
        hipLaunchKernelGGL((cell_table_grad_kernel<5, false, float>), tgrid, tblock, G::lds_bytes(), st, *plan, h, L, pbuf, plane, grad_out, grad_table_v);
        hipLaunchKernelGGL((cell_table_grad_kernel<5, true, T>), tgrid, tblock, G::lds_bytes(), st, *plan, h, L, gsbuf, plane, k, grad_table_k);
    }
    check_launch();
}

}  // namespace p2

using namespace p2;

extern "C" {

void cell_attention_forward_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *q, const float *k,
                                     const float *v, const float *table_q, const float *table_k, const float *table_v, float *out,
                                     float *ml, float *pbuf) {
    launch_cell_fwd<float>(plan, h, hdim, L, q, k, v, table_q, table_k, table_v, out, ml, pbuf);
}
void cell_attention_backward_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *grad_out, const float *q,
                                      const float *k, const float *v, const float *out, const float *table_q, const float *table_k,
                                      const float *table_v, const float *pbuf, float *gsbuf, float *grad_q, float *grad_k,
                                      float *grad_v, float *grad_table_q, float *grad_table_k, float *grad_table_v) {
    launch_cell_bwd<float>(plan, h, hdim, L, grad_out, q, k, v, out, table_q, table_k, table_v, pbuf, gsbuf, grad_q, grad_k, grad_v, grad_table_q,
                           grad_table_k, grad_table_v);
}
// bf16 storage of q / k / v / tables (raw 16-bit patterns), fp32 arithmetic, fp32 outputs and gradients
void cell_attention_forward_bf16_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const uint16_t *q, const uint16_t *k,
                                          const uint16_t *v, const uint16_t *table_q, const uint16_t *table_k, const uint16_t *table_v,
                                          float *out, float *ml, float *pbuf) {
    launch_cell_fwd<bf16_t>(plan, h, hdim, L, q, k, v, table_q, table_k, table_v, out, ml, pbuf);
}
void cell_attention_backward_bf16_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *grad_out, const uint16_t *q,
                                           const uint16_t *k, const uint16_t *v, const float *out, const uint16_t *table_q,
                                           const uint16_t *table_k, const uint16_t *table_v, const float *pbuf, float *gsbuf, float *grad_q,
                                           float *grad_k, float *grad_v, float *grad_table_q, float *grad_table_k, float *grad_table_v) {
    launch_cell_bwd<bf16_t>(plan, h, hdim, L, grad_out, q, k, v, out, table_q, table_k, table_v, pbuf, gsbuf, grad_q, grad_k, grad_v, grad_table_q,
                            grad_table_k, grad_table_v);
}

}  // extern "C"
