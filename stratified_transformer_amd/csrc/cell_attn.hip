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
#include "rpe_common.h"

namespace p2 {

typedef float __attribute__((ext_vector_type(4))) f32x4c;

constexpr int CA_NP = 8;      // passes of 16 keys a lane keeps in registers: 128 keys per chunk
constexpr int CA_WAVES = 12;  // waves per workgroup (one head's three tables in LDS per workgroup)

// ---- cross-lane sums without LDS round trips where the hardware has a lane network for it ----
template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float swap16(float v) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)); }
// over the four quarter lanes of a key slot (lane bits 0-1): quad_perm [1,0,3,2], [2,3,0,1]
__device__ __forceinline__ float quad_sum(float v) {
    v += dppf<0xB1>(v);
    v += dppf<0x4E>(v);
    return v;
}
// over the 16 key slots (lane bits 2-5), every lane gets the result: row_ror:4, row_ror:8, swap of 16-lane rows, xor 32
__device__ __forceinline__ float slots_sum16(float v) {
    v += dppf<0x124>(v);
    v += dppf<0x128>(v);
    v += swap16(v);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float slots_max16(float v) {
    v = fmaxf(v, dppf<0x124>(v));
    v = fmaxf(v, dppf<0x128>(v));
    v = fmaxf(v, swap16(v));
    v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}
__device__ __forceinline__ float4 slots_sum16_4(float4 v) {
    return make_float4(slots_sum16(v.x), slots_sum16(v.y), slots_sum16(v.z), slots_sum16(v.w));
}

// ---- buffer addressing: a wave-uniform base in scalar registers + one 32-bit byte offset per lane; the passes of a
// row differ by an immediate.  Reads past the end of the buffer return 0, stores past it are dropped. ----
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ unsigned bload_u32(rsrc_t r, int off) { return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0); }
__device__ __forceinline__ float bload_f32(rsrc_t r, int off) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0)); }
__device__ __forceinline__ void bstore_f32(rsrc_t r, int off, float v) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, 0, 0); }

// ---- LDS image of a head's three tables: table TB at float offset TB * TS (TS a compile-time constant, so that the
// three tables of one (axis, row) differ by an immediate offset), inside a table [axis][row][16] ----
template <int LCAP>
struct TabGeo {
    static constexpr int TS = 3 * LCAP * 16;  // floats per table image
    static constexpr size_t bytes() { return (size_t)3 * TS * sizeof(float); }
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

struct CellTask {
    int qs, nq, kb, nk, pbase;
};
// (everything about a cell is wave-uniform: held in scalar registers)
__device__ __forceinline__ CellTask cell_task(const pointops2_cell_plan &pl, int task) {
    const int cell = __builtin_amdgcn_readfirstlane(pl.cell_perm[task]);
    CellTask t;
    t.qs = __builtin_amdgcn_readfirstlane(pl.cell_qstart[cell]);
    t.nq = __builtin_amdgcn_readfirstlane(pl.cell_qstart[cell + 1]) - t.qs;
    t.kb = __builtin_amdgcn_readfirstlane(pl.cell_kbase[cell]);
    t.nk = __builtin_amdgcn_readfirstlane(pl.cell_kbase[cell + 1]) - t.kb;
    t.pbase = __builtin_amdgcn_readfirstlane(pl.cell_pbase[cell]);
    return t;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int NP, int LCAP>
__global__ __launch_bounds__(CA_WAVES * 64) void cell_fwd_kernel(pointops2_cell_plan pl, int h, int L, const float *__restrict__ q,
                                                                 const float *__restrict__ k, const float *__restrict__ v,
                                                                 const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                                 const float *__restrict__ table_v, float *__restrict__ out,
                                                                 float *__restrict__ ml, float *__restrict__ pbuf, size_t plane) {
    constexpr int D = 16, TS = TabGeo<LCAP>::TS;
    extern __shared__ float lds[];
    const int head = blockIdx.y, C = h * D;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, p = lane >> 2, c = lane & 3;
    stage_table<D>(lds, table_q, L, h, head, 1);
    stage_table<D>(lds + TS, table_k, L, h, head, 1);
    stage_table<D>(lds + 2 * TS, table_v, L, h, head, 1);
    __syncthreads();
    const int nC = pl.counts[0];
    float *pb = pbuf + (size_t)head * plane;
    const int hoff = head * D + 4 * c;
    for (int task = blockIdx.x * CA_WAVES + wave; task < nC; task += gridDim.x * CA_WAVES) {
        const CellTask ct = cell_task(pl, task);
        const int nch = (ct.nk + 16 * NP - 1) / (16 * NP);
        const unsigned tile_bytes = (unsigned)ct.nq * ct.nk * 4u;
        const rsrc_t rs_rel = make_rsrc(pl.relp + ct.pbase, tile_bytes);
        const rsrc_t rs_p = make_rsrc(pb + ct.pbase, tile_bytes);
        const rsrc_t rs_key = make_rsrc(pl.cell_keys + ct.kb, (unsigned)ct.nk * 4u);
        const rsrc_t rs_qid = make_rsrc(pl.cell_order + ct.qs, (unsigned)ct.nq * 4u);
        // ---- sweep 1: logits and softmax ----
        for (int ch = 0; ch < nch; ch++) {
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0), np = (nkc + 15) >> 4;
            float4 k4[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                const int key = (int)bload_u32(rs_key, (j0 + 16 * t + p) * 4);  // (past the end: key 0, never used)
                k4[t] = ldg4(k + (size_t)key * C + hoff);
            }
            // the inputs of query il+1 are requested while query il is worked on
            int i_nx = (int)bload_u32(rs_qid, 0);
            float4 q4_nx = ldg4(q + (size_t)i_nx * C + hoff);
            unsigned w_nx[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) w_nx[t] = bload_u32(rs_rel, (j0 + p) * 4 + 64 * t);
            for (int il = 0; il < ct.nq; il++) {
                const int i = i_nx;
                const float4 q4 = q4_nx;
                unsigned w[NP];
#pragma unroll
                for (int t = 0; t < NP; t++) w[t] = w_nx[t];
                const int roff = (il * ct.nk + j0 + p) * 4;
                {
                    i_nx = (int)bload_u32(rs_qid, (il + 1) * 4);  // (past the end: query 0, never used)
                    q4_nx = ldg4(q + (size_t)i_nx * C + hoff);
#pragma unroll
                    for (int t = 0; t < NP; t++)
                        if (t < np) w_nx[t] = bload_u32(rs_rel, roff + ct.nk * 4 + 64 * t);
                }
                float lg[NP];
                float mx = -INFINITY;
#pragma unroll
                for (int t = 0; t < NP; t++) {
                    lg[t] = -INFINITY;
                    if (t < np) {  // wave-uniform
                        const RowOff ro = row_off(w[t], L, c);
                        const float s = quad_sum(dot4(q4, k4[t]) + dot4(q4, tsum_at<0>(lds, ro)) + dot4(k4[t], tsum_at<TS>(lds, ro)));
                        if (16 * t + p < nkc && !(w[t] >> 31)) lg[t] = s;
                        mx = fmaxf(mx, lg[t]);
                    }
                }
                mx = slots_max16(mx);
                if (nch == 1) {
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < NP; t++)
                        if (t < np) {
                            lg[t] = __expf(lg[t] - mx);  // exp(-inf) = 0: masked entries and slots past the end
                            sum += lg[t];
                        }
                    const float inv = 1.0f / slots_sum16(sum);
#pragma unroll
                    for (int t = 0; t < NP; t++)
                        if (t < np && c == 0 && 16 * t + p < nkc) bstore_f32(rs_p, roff + 64 * t, lg[t] * inv);
                } else {
                    float *st = ml + ((size_t)i * h + head) * 2;
                    const float m_old = ch ? st[0] : -INFINITY, l_old = ch ? st[1] : 0.f;
                    const float m_new = fmaxf(m_old, mx);
                    float sum = 0.f;
#pragma unroll
                    for (int t = 0; t < NP; t++)
                        if (t < np) {
                            sum += __expf(lg[t] - m_new);
                            if (c == 0 && 16 * t + p < nkc) bstore_f32(rs_p, roff + 64 * t, lg[t]);  // logits; sweep 2 makes them weights
                        }
                    sum = slots_sum16(sum);
                    if (lane == 0) {
                        st[0] = m_new;
                        st[1] = (m_old == -INFINITY ? 0.f : l_old * __expf(m_old - m_new)) + sum;
                    }
                }
            }
        }
        // ---- sweep 2: out = sum p (v + Tv) ----
        for (int ch = 0; ch < nch; ch++) {
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0), np = (nkc + 15) >> 4;
            float4 v4[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                const int key = (int)bload_u32(rs_key, (j0 + 16 * t + p) * 4);
                v4[t] = ldg4(v + (size_t)key * C + hoff);
            }
            unsigned w_nx[NP];
            float a_nx[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                w_nx[t] = bload_u32(rs_rel, (j0 + p) * 4 + 64 * t);
                a_nx[t] = bload_f32(rs_p, (j0 + p) * 4 + 64 * t);
            }
            int i_nx = (int)bload_u32(rs_qid, 0);
            for (int il = 0; il < ct.nq; il++) {
                const int i = i_nx;
                const int roff = (il * ct.nk + j0 + p) * 4;
                unsigned w[NP];
                float a[NP];
#pragma unroll
                for (int t = 0; t < NP; t++) {
                    w[t] = w_nx[t];
                    a[t] = a_nx[t];
                }
                i_nx = (int)bload_u32(rs_qid, (il + 1) * 4);
#pragma unroll
                for (int t = 0; t < NP; t++)
                    if (t < np) {
                        w_nx[t] = bload_u32(rs_rel, roff + ct.nk * 4 + 64 * t);
                        a_nx[t] = bload_f32(rs_p, roff + ct.nk * 4 + 64 * t);
                    }
                if (nch > 1) {
                    const float *st = ml + ((size_t)i * h + head) * 2;
                    const float m = st[0], inv = 1.0f / st[1];
#pragma unroll
                    for (int t = 0; t < NP; t++)
                        if (t < np) {
                            a[t] = __expf(a[t] - m) * inv;
                            if (c == 0 && 16 * t + p < nkc) bstore_f32(rs_p, roff + 64 * t, a[t]);
                        }
                }
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < NP; t++)
                    if (t < np) {
                        const RowOff ro = row_off(w[t], L, c);
                        const float at = 16 * t + p < nkc ? a[t] : 0.f;  // (a masked entry's weight is stored as 0)
                        acc = fma4(at, add4(tsum_at<2 * TS>(lds, ro), v4[t]), acc);
                    }
                const float4 tot = slots_sum16_4(acc);
                if (p == 0) {
                    float *o = out + (size_t)i * C + hoff;
                    stg4(o, ch ? add4(tot, ldg4(o)) : tot);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, sweeps A and B
// ------------------------------------------------------------------------------------------------
// adds the key-side accumulators of one pass (lane (p, c): floats 4c..4c+3 of key slot p) to grad[key, head, :]:
// through a wave-private LDS tile, so that one atomic instruction covers four whole 64-byte head rows
__device__ __forceinline__ void flush_key_pass(float *scr, float4 acc, rsrc_t rs_key, int jbase, int nleft, float *__restrict__ grad,
                                               int C, int head, int lane, int p, int c) {
    *reinterpret_cast<float4 *>(scr + p * 16 + 4 * c) = acc;
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        const int s = (lane >> 4) + 4 * kk;
        const int key = (int)bload_u32(rs_key, (jbase + s) * 4);
        if (s < nleft) unsafeAtomicAdd(grad + (size_t)key * C + head * 16 + (lane & 15), scr[s * 16 + (lane & 15)]);
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_wave_barrier();
}

template <int NP, int LCAP>
__global__ __launch_bounds__(CA_WAVES * 64) void cell_bwd_kernel(pointops2_cell_plan pl, int h, int L, const float *__restrict__ go,
                                                                 const float *__restrict__ q, const float *__restrict__ k,
                                                                 const float *__restrict__ v, const float *__restrict__ out,
                                                                 const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                                 const float *__restrict__ table_v, const float *__restrict__ pbuf,
                                                                 float *__restrict__ gsbuf, size_t plane, float *__restrict__ grad_q,
                                                                 float *__restrict__ grad_k, float *__restrict__ grad_v) {
    constexpr int D = 16, TS = TabGeo<LCAP>::TS;
    extern __shared__ float lds[];
    const int head = blockIdx.y, C = h * D;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, p = lane >> 2, c = lane & 3;
    float *scr = lds + 3 * TS + wave * 256;
    stage_table<D>(lds, table_q, L, h, head, 1);
    stage_table<D>(lds + TS, table_k, L, h, head, 1);
    stage_table<D>(lds + 2 * TS, table_v, L, h, head, 1);
    __syncthreads();
    const int nC = pl.counts[0];
    const float *pb = pbuf + (size_t)head * plane;
    float *gb = gsbuf + (size_t)head * plane;
    const int hoff = head * D + 4 * c;
    for (int task = blockIdx.x * CA_WAVES + wave; task < nC; task += gridDim.x * CA_WAVES) {
        const CellTask ct = cell_task(pl, task);
        const int nch = (ct.nk + 16 * NP - 1) / (16 * NP);
        const unsigned tile_bytes = (unsigned)ct.nq * ct.nk * 4u;
        const rsrc_t rs_rel = make_rsrc(pl.relp + ct.pbase, tile_bytes);
        const rsrc_t rs_p = make_rsrc(pb + ct.pbase, tile_bytes);
        const rsrc_t rs_g = make_rsrc(gb + ct.pbase, tile_bytes);
        const rsrc_t rs_key = make_rsrc(pl.cell_keys + ct.kb, (unsigned)ct.nk * 4u);
        const rsrc_t rs_qid = make_rsrc(pl.cell_order + ct.qs, (unsigned)ct.nq * 4u);
        // ---- sweep A: grad_attn = <go, v + Tv>, gs = p (grad_attn - <go, out>), dV += p go ----
        for (int ch = 0; ch < nch; ch++) {
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0), np = (nkc + 15) >> 4;
            float4 v4[NP], dv4[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                const int key = (int)bload_u32(rs_key, (j0 + 16 * t + p) * 4);
                v4[t] = ldg4(v + (size_t)key * C + hoff);
                dv4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            unsigned w_nx[NP];
            float a_nx[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                w_nx[t] = bload_u32(rs_rel, (j0 + p) * 4 + 64 * t);
                a_nx[t] = bload_f32(rs_p, (j0 + p) * 4 + 64 * t);
            }
            int i_nx = (int)bload_u32(rs_qid, 0);
            float4 go_nx = ldg4(go + (size_t)i_nx * C + hoff), o_nx = ldg4(out + (size_t)i_nx * C + hoff);
            for (int il = 0; il < ct.nq; il++) {
                const int roff = (il * ct.nk + j0 + p) * 4;
                const float4 go4 = go_nx, o4 = o_nx;
                unsigned w[NP];
                float a[NP];
#pragma unroll
                for (int t = 0; t < NP; t++) {
                    w[t] = w_nx[t];
                    a[t] = a_nx[t];
                }
                i_nx = (int)bload_u32(rs_qid, (il + 1) * 4);
                go_nx = ldg4(go + (size_t)i_nx * C + hoff);
                o_nx = ldg4(out + (size_t)i_nx * C + hoff);
#pragma unroll
                for (int t = 0; t < NP; t++)
                    if (t < np) {
                        w_nx[t] = bload_u32(rs_rel, roff + ct.nk * 4 + 64 * t);
                        a_nx[t] = bload_f32(rs_p, roff + ct.nk * 4 + 64 * t);
                    }
                const float delta = quad_sum(dot4(go4, o4));  // = sum over the row of p * grad_attn
#pragma unroll
                for (int t = 0; t < NP; t++)
                    if (t < np) {
                        const RowOff ro = row_off(w[t], L, c);
                        const float at = 16 * t + p < nkc ? a[t] : 0.f;
                        const float ga = quad_sum(dot4(go4, add4(tsum_at<2 * TS>(lds, ro), v4[t])));
                        if (c == 0 && 16 * t + p < nkc) bstore_f32(rs_g, roff + 64 * t, at * (ga - delta));
                        dv4[t] = fma4(at, go4, dv4[t]);
                    }
            }
#pragma unroll
            for (int t = 0; t < NP; t++)
                if (t < np) flush_key_pass(scr, dv4[t], rs_key, j0 + 16 * t, nkc - 16 * t, grad_v, C, head, lane, p, c);
        }
        // ---- sweep B: dQ = sum gs (k + Tq), dK += gs (q + Tk) ----
        for (int ch = 0; ch < nch; ch++) {
            const int j0 = ch * 16 * NP, nkc = min(16 * NP, ct.nk - j0), np = (nkc + 15) >> 4;
            float4 k4[NP], dk4[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                const int key = (int)bload_u32(rs_key, (j0 + 16 * t + p) * 4);
                k4[t] = ldg4(k + (size_t)key * C + hoff);
                dk4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            unsigned w_nx[NP];
            float g_nx[NP];
#pragma unroll
            for (int t = 0; t < NP; t++) {
                w_nx[t] = bload_u32(rs_rel, (j0 + p) * 4 + 64 * t);
                g_nx[t] = bload_f32(rs_g, (j0 + p) * 4 + 64 * t);
            }
            int i_nx = (int)bload_u32(rs_qid, 0);
            float4 q_nx = ldg4(q + (size_t)i_nx * C + hoff);
            for (int il = 0; il < ct.nq; il++) {
                const int i = i_nx;
                const int roff = (il * ct.nk + j0 + p) * 4;
                const float4 q4 = q_nx;
                unsigned w[NP];
                float g[NP];
#pragma unroll
                for (int t = 0; t < NP; t++) {
                    w[t] = w_nx[t];
                    g[t] = g_nx[t];
                }
                i_nx = (int)bload_u32(rs_qid, (il + 1) * 4);
                q_nx = ldg4(q + (size_t)i_nx * C + hoff);
#pragma unroll
                for (int t = 0; t < NP; t++)
                    if (t < np) {
                        w_nx[t] = bload_u32(rs_rel, roff + ct.nk * 4 + 64 * t);
                        g_nx[t] = bload_f32(rs_g, roff + ct.nk * 4 + 64 * t);
                    }
                float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < NP; t++)
                    if (t < np) {
                        const RowOff ro = row_off(w[t], L, c);
                        const float gt = 16 * t + p < nkc ? g[t] : 0.f;
                        dq = fma4(gt, add4(tsum_at<0>(lds, ro), k4[t]), dq);
                        dk4[t] = fma4(gt, add4(tsum_at<TS>(lds, ro), q4), dk4[t]);
                    }
                const float4 tot = slots_sum16_4(dq);
                if (p == 0) {
                    float *o = grad_q + (size_t)i * C + hoff;
                    stg4(o, ch ? add4(tot, ldg4(o)) : tot);
                }
            }
#pragma unroll
            for (int t = 0; t < NP; t++)
                if (t < np) flush_key_pass(scr, dk4[t], rs_key, j0 + 16 * t, nkc - 16 * t, grad_k, C, head, lane, p, c);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// table gradients on cell rows (the scheme of rpe_bwd_mfma.hip's table_grad_kernel, see its header):
//   BYKEY = false: rows = sorted query positions; a row is the query's contiguous row of its cell's tile; X = q or grad_out
//   BYKEY = true:  rows = key slots; a row is the key's column of the tile (stride n_k); X = k
// grad_table[r, head, i, ax] += sum over rows, over the row's entries with rel[ax] == r, of w * X[row point, head, i]
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

constexpr int CT_WAVES = 12;  // rows per group = K of the outer-product step (3 x 4)
constexpr int CT_MAXP = 8;    // passes of 16 entries a row segment is walked from registers

template <int TA>
struct CellTableGeo {
    static constexpr int LP = TA * 16;       // padded bins per axis
    static constexpr int ROW = 3 * LP + 16;  // ints per histogram row (+16: the four k-rows of one ds_read on different banks)
    static constexpr size_t walk_bytes() { return (size_t)CT_WAVES * (ROW + 16) * 4; }
    static constexpr size_t flush_bytes() { return (size_t)CT_WAVES * 16 * 48 * 4; }
    static constexpr size_t lds_bytes() { return walk_bytes() > flush_bytes() ? walk_bytes() : flush_bytes(); }
};

template <int TA, bool BYKEY>
__global__ __launch_bounds__(CT_WAVES * 64) void cell_table_grad_kernel(pointops2_cell_plan pl, int nrows_fixed, int h, int L,
                                                                        const float *__restrict__ wbuf, size_t plane,
                                                                        const float *__restrict__ X, float *__restrict__ grad_table) {
    constexpr int D = 16, NW = CT_WAVES;
    using G = CellTableGeo<TA>;
    extern __shared__ float lds[];
    int *hist = reinterpret_cast<int *>(lds);  // [NW][ROW]
    float *xs = lds + NW * G::ROW;             // [NW][16]: X rows of the group, scaled by 2^-S
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int C = h * D;
    const int p = lane >> 2, c = lane & 3;
    const int head = blockIdx.y;
    const int kq = lane >> 4, col = lane & 15;
    const float *wb = wbuf + (size_t)head * plane;
    const int N = nrows_fixed;  // rows: key slots | queries

    f32x4c acc[3];
#pragma unroll
    for (int ax = 0; ax < 3; ax++) acc[ax] = f32x4c{0.f, 0.f, 0.f, 0.f};
    for (int x = threadIdx.x; x < NW * G::ROW; x += NW * 64) hist[x] = 0;
    __shared__ int next_row;
    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int rb = min(N, (int)blockIdx.x * per), re = min(N, rb + per);
    if (threadIdx.x == 0) next_row = rb;
    __syncthreads();
    int *myh = hist + wave * G::ROW;
    auto claim = [&]() -> int {
        int r = 0;
        if (lane == 0) r = atomicAdd(&next_row, 1);
        r = __builtin_amdgcn_readfirstlane(r);
        return r < re ? r : -1;
    };
    // a row = (first entry, entries, stride between entries, point whose X row it multiplies)
    struct Row {
        size_t start;
        int n, stride, point;
    };
    auto describe = [&](int r) -> Row {
        Row d;
        if (BYKEY) {
            const int cell = pl.kcell[r];
            const int kb = pl.cell_kbase[cell];
            d.stride = pl.cell_kbase[cell + 1] - kb;
            d.start = (size_t)pl.cell_pbase[cell] + (r - kb);
            d.n = pl.cell_qstart[cell + 1] - pl.cell_qstart[cell];
            d.point = pl.cell_keys[r];
        } else {
            const int cell = pl.qcell[r];
            const int nk = pl.cell_kbase[cell + 1] - pl.cell_kbase[cell];
            d.stride = 1;
            d.start = (size_t)pl.cell_pbase[cell] + (size_t)(r - pl.cell_qstart[cell]) * nk;
            d.n = nk;
            d.point = pl.cell_order[r];
        }
        return d;
    };
    int row = -1, cur = 0;
    Row rd{0, 0, 1, 0}, nrd{0, 0, 1, 0};
    int nrow = claim();
    if (nrow >= 0) nrd = describe(nrow);
    for (;;) {
        if (cur >= rd.n || row < 0) {  // this wave's row is finished: take the claimed one (none left: row = -1 from here on)
            row = nrow;
            rd = nrd;
            cur = 0;
            nrow = -1;
            if (row < 0) rd.n = 0;
        }
        const bool last_segment = row >= 0 && rd.n - cur <= 16 * CT_MAXP;
        if (row >= 0) {
            const int s = cur, e = min(rd.n, cur + 16 * CT_MAXP);
            cur = e;
            const int np = (e - s + 15) >> 4;
            unsigned rreg[CT_MAXP];
            float wreg[CT_MAXP];
#pragma unroll
            for (int i = 0; i < CT_MAXP; i++) { rreg[i] = 0; wreg[i] = 0.f; }
            const float4 x4 = ldg4(X + (size_t)rd.point * C + head * D + 4 * c);
            if (e > s) {
#pragma unroll
                for (int i = 0; i < CT_MAXP; i++) {
                    const size_t pos = rd.start + (size_t)min(s + i * 16 + p, e - 1) * rd.stride;
                    rreg[i] = pl.relp[pos];
                    wreg[i] = wb[pos];
                }
            }
            if (last_segment) {  // wave-uniform; behind this segment's loads so that the claim's round trips overlap them
                nrow = claim();
                if (nrow >= 0) nrd = describe(nrow);
            }
            unsigned mxb = 0u;
#pragma unroll
            for (int i = 0; i < CT_MAXP; i++) {
                const bool live = s + i * 16 + p < e && !(rreg[i] >> 31);
                wreg[i] = live ? wreg[i] : 0.f;
                mxb = max(mxb, __float_as_uint(fabsf(wreg[i])));
            }
            const CFixScale sc = c_row_scale(wave_max_u32(mxb), e - s);
            if (lane < 4)
                *reinterpret_cast<float4 *>(&xs[wave * 16 + 4 * c]) = make_float4(x4.x * sc.inv, x4.y * sc.inv, x4.z * sc.inv, x4.w * sc.inv);
#pragma unroll
            for (int i = 0; i < CT_MAXP; i++) {
                if (i < np) {  // wave-uniform
                    const int r = (rreg[i] >> (8 * min(c, 2))) & 255;
                    if (wreg[i] != 0.f && c < 3) atomicAdd(&myh[c * G::LP + r], __float2int_rn(wreg[i] * sc.mul));
                }
            }
        } else if (lane < 4) {
            *reinterpret_cast<float4 *>(&xs[wave * 16 + 4 * c]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const int more = __syncthreads_or(row >= 0);
        if (!more) break;
        if (wave < TA) {  // wave bt owns the three axis tiles of bin tile bt
            const int bt = wave;
#pragma unroll
            for (int ks = 0; ks < NW / 4; ks++) {
                const int rk = ks * 4 + kq;
                const float b = xs[rk * 16 + col];
#pragma unroll
                for (int ax = 0; ax < 3; ax++) {
                    int *hp = &hist[rk * G::ROW + ax * G::LP + bt * 16 + col];
                    const float a = (float)*hp;
                    *hp = 0;
                    acc[ax] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[ax], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // flush: the wave's (bin tile, 3 axes) goes through LDS in table order ([L, h, 16, 3]: the 48 floats of one (bin, head)
    // are contiguous) and leaves as 12 instructions of 64 consecutive floats
    __syncthreads();
    if (wave < TA) {
        float *stage = lds + wave * (16 * 48);
        const int bt = wave;
#pragma unroll
        for (int ax = 0; ax < 3; ax++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) stage[((kq * 4 + reg) * 16 + col) * 3 + ax] = acc[ax][reg];
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int x = lane; x < 16 * 48; x += 64) {
            const int bin = bt * 16 + x / 48;
            const float val = stage[x];
            if (bin < L && val != 0.f) unsafeAtomicAdd(grad_table + ((size_t)bin * h + head) * 48 + x % 48, val);
        }
    }
}

static int cell_grid_x(int N, int h) {
    // persistent grid: two workgroups per CU over all heads, never more waves than a generous bound on the cells
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = kNumCU;
        return n;
    }();
    const int cap = max(1, 2 * cus / max(h, 1));
    return max(1, min(cap, div_up(N, CA_WAVES)));
}

}  // namespace p2

using namespace p2;

extern "C" {

void cell_attention_forward_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *q, const float *k,
                                     const float *v, const float *table_q, const float *table_k, const float *table_v, float *out,
                                     float *ml, float *pbuf) {
    if (plan == nullptr || plan->n_points <= 0) return;
    if (hdim != 16) { set_error("cell_attention: d != 16"); return; }
    if (L < 1) { set_error("cell_attention: no table rows"); return; }
    const dim3 grid(cell_grid_x(plan->n_points, h), h), block(CA_WAVES * 64);
    const size_t plane = (size_t)plan->n_pairs;
    if (L <= 80) {
        const size_t lds = TabGeo<80>::bytes();
        allow_big_lds(cell_fwd_kernel<CA_NP, 80>, lds);
        hipLaunchKernelGGL((cell_fwd_kernel<CA_NP, 80>), grid, block, lds, state().stream, *plan, h, L, q, k, v, table_q, table_k, table_v, out, ml,
                           pbuf, plane);
    } else if (L <= 160) {
        const size_t lds = TabGeo<160>::bytes();
        allow_big_lds(cell_fwd_kernel<CA_NP, 160>, lds);
        hipLaunchKernelGGL((cell_fwd_kernel<CA_NP, 160>), grid, block, lds, state().stream, *plan, h, L, q, k, v, table_q, table_k, table_v, out, ml,
                           pbuf, plane);
    } else {
        set_error("cell_attention: more than 160 table rows (use the operators)");
        return;
    }
    check_launch();
}

void cell_attention_backward_launcher(const pointops2_cell_plan *plan, int h, int hdim, int L, const float *grad_out, const float *q,
                                      const float *k, const float *v, const float *out, const float *table_q, const float *table_k,
                                      const float *table_v, const float *pbuf, float *gsbuf, float *grad_q, float *grad_k,
                                      float *grad_v, float *grad_table_q, float *grad_table_k, float *grad_table_v) {
    if (plan == nullptr || plan->n_points <= 0) return;
    if (hdim != 16) { set_error("cell_attention: d != 16"); return; }
    if (L < 1 || L > 80) { set_error("cell_attention backward: table rows L must be in 1..80"); return; }
    hipStream_t st = state().stream;
    const size_t lds = TabGeo<80>::bytes() + (size_t)CA_WAVES * 256 * sizeof(float);
    allow_big_lds(cell_bwd_kernel<CA_NP, 80>, lds);
    const int N = plan->n_points;
    const size_t plane = (size_t)plan->n_pairs;
    hipLaunchKernelGGL((cell_bwd_kernel<CA_NP, 80>), dim3(cell_grid_x(N, h), h), dim3(CA_WAVES * 64), lds, st, *plan, h, L, grad_out, q, k, v, out,
                       table_q, table_k, table_v, pbuf, gsbuf, plane, grad_q, grad_k, grad_v);
    // the three table gradients read p / gs only
    const int gx_q = max(1, min(cell_grid_x(N, h) * 2, div_up(N, CT_WAVES)));
    const int gx_k = max(1, min(cell_grid_x(N, h) * 2, div_up(plan->n_keyslots, CT_WAVES)));
    if (L <= 64) {
        using G = CellTableGeo<4>;
        hipLaunchKernelGGL((cell_table_grad_kernel<4, false>), dim3(gx_q, h), dim3(CT_WAVES * 64), G::lds_bytes(), st, *plan, N, h, L, gsbuf, plane,
                           q, grad_table_q);
        hipLaunchKernelGGL((cell_table_grad_kernel<4, false>), dim3(gx_q, h), dim3(CT_WAVES * 64), G::lds_bytes(), st, *plan, N, h, L, pbuf, plane,
                           grad_out, grad_table_v);
        hipLaunchKernelGGL((cell_table_grad_kernel<4, true>), dim3(gx_k, h), dim3(CT_WAVES * 64), G::lds_bytes(), st, *plan, plan->n_keyslots, h, L,
                           gsbuf, plane, k, grad_table_k);
    } else {
        using G = CellTableGeo<5>;
        hipLaunchKernelGGL((cell_table_grad_kernel<5, false>), dim3(gx_q, h), dim3(CT_WAVES * 64), G::lds_bytes(), st, *plan, N, h, L, gsbuf, plane,
                           q, grad_table_q);
        hipLaunchKernelGGL((cell_table_grad_kernel<5, false>), dim3(gx_q, h), dim3(CT_WAVES * 64), G::lds_bytes(), st, *plan, N, h, L, pbuf, plane,
                           grad_out, grad_table_v);
        hipLaunchKernelGGL((cell_table_grad_kernel<5, true>), dim3(gx_k, h), dim3(CT_WAVES * 64), G::lds_bytes(), st, *plan, plan->n_keyslots, h, L,
                           gsbuf, plane, k, grad_table_k);
    }
    check_launch();
}

}  // extern "C"
