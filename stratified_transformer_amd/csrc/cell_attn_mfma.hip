// Window-centric ("cell") attention FORWARD on the matrix cores, gfx950.
//
// cell_attn.hip's forward evaluates every (query, key) pair of a cell's dense n_q x n_k tile on the VALU and reads NINE
// 64-byte table rows per (pair, head) from LDS (Tq, Tk, Tv x 3 axes): it is bound by VALU issue and LDS row reads, at ~4 %
// of the fp32 peak with zero MFMA instructions.  Here the three dot products of a pair are factored so that everything that
// has a matrix shape runs as v_mfma_f32_16x16x4_f32 (exact fp32: an fma chain) and a pair costs SCALAR LDS reads only:
//
//   logit(i, j) = <q_i, k_j> + sum_ax QT[i][ax][r_ax(i,j)] + sum_ax KT[j][ax][r_ax(i,j)]
//       S  = K Q^T                 16 keys x 16 queries x 16 features     = 4 MFMAs per (key tile, query tile)
//       QT = Tq_ax Q^T             LP rows  x 16 queries                  = LP/4 MFMAs per (axis, query tile), once per piece
//       KT = Tk_ax K^T             LP rows  x 16 keys                     = LP/4 MFMAs per (axis, key tile)
//     QT / KT tiles go to a per-wave LDS buffer ([16][LP+4] floats per axis) and a pair reads ONE float of each
//     (model/stratified_transformer.py:194 -> relative_pos_encoding_cuda_kernel_v2.cu:276-279 reads 6 table rows per pair)
//   out(i)  = sum_j p_ij v_j  +  sum_ax sum_r H[i][ax][r] Tv[r,:,ax],     H[i][ax][r] = sum of p_ij over the pairs with r_ax = r
//       P V                        4 MFMAs per (key tile, query tile): the softmax weights ARE the B operand as they stand in
//                                  the accumulator layout of S (the contraction runs over a permutation of the tile's keys)
//       H  histogram               3 integer LDS atomics per pair (2^30 fixed point: p <= 1, a row's weights sum to 1; LDS float
//                                  atomics are ~10x slower on gfx950)
//       H Tv                       LP/4 MFMAs per (axis, query tile), once per piece (:208 -> ...kernel_v2.cu:430 reads 3 rows per pair)
//
// One wave per (cell piece, head), as in cell_attn.hip (same plan, same task dealing, same pbuf / out contract: pbuf receives
// the softmax weights in tile order, flagged and out-of-tile entries are never read back).  The logits of a chunk of up to
// 16 * CM_NKT keys x 16 * NQT queries live in registers in the MFMA accumulator layout; longer key lists run in chunks
// (raw logits parked in pbuf, running max / sum in registers, weights made in a second sweep).
// Tables: one head's three tables are staged per workgroup in MFMA FRAGMENT order (one ds_read_b128 = the A operand of the
// four k-steps of a 16-row tile), rows padded with zeros to LP = 64 or 80 per axis.
#include "cell_common.h"

namespace p2 {

constexpr int CM_NKT = 8;  // key tiles (16 keys each) of a chunk whose logits stay in registers

template <int LP>
struct CmGeo {
    static constexpr int NTA = LP / 16;              // 16-row tiles per axis
    static constexpr int RS = LP + 4;                // floats per row of a lookup tile (16-byte aligned rows, banks spread by 4)
    static constexpr int TILE = 16 * RS;             // floats per lookup tile: 16 queries (or keys) x one axis
    static constexpr int TAB = 3 * NTA * 256;        // floats per table image in fragment order
    static constexpr int WAVES = LP <= 64 ? 12 : 9;  // waves per workgroup (LDS: 3 tables + 2 lookup tiles per wave; <= 168 registers)
    static constexpr int WAVE_FLOATS = 2 * TILE;
    static constexpr size_t lds_bytes() { return (size_t)(3 * TAB + WAVES * WAVE_FLOATS) * 4; }
};

__device__ __forceinline__ f32x4c mfma4(float4 a, float4 b, f32x4c c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ void lds_fence() {
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): this wave's LDS writes / atomics have landed
    __builtin_amdgcn_wave_barrier();
}
// max / sum over the four lanes n, n + 16, n + 32, n + 48 (the four k-groups of one query column)
__device__ __forceinline__ float col_max(float v) {
    v = fmaxf(v, swap16(v));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float col_sum(float v) {
    v += swap16(v);
    return v + __shfl_xor(v, 32, 64);
}

// fragment-ordered image of one table (global layout [L, h, 16, 3]) for one head:
//   ROWS_IN_M = true   A operand of  D[row][x] = sum_feat T[row][feat] X[x][feat]:  img[(ax*NTA+T)*256 + lane*4 + s] = T[16T + (lane&15)][4(lane>>4) + s]
//   ROWS_IN_M = false  A operand of  D[feat][x] = sum_row T[row][feat] H[x][row]:    img[(ax*NTA+T)*256 + lane*4 + s] = T[16T + 4(lane>>4) + s][lane&15]
template <int LP, bool ROWS_IN_M>
__device__ __forceinline__ void stage_fragments(float *img, const float *__restrict__ tab, int L, int h, int head) {
    constexpr int NTA = LP / 16;
    for (int x = threadIdx.x; x < 3 * NTA * 256; x += blockDim.x) {
        const int s = x & 3, ln = (x >> 2) & 63, T = (x >> 8) % NTA, ax = (x >> 8) / NTA;
        const int r = ROWS_IN_M ? 16 * T + (ln & 15) : 16 * T + 4 * (ln >> 4) + s;
        const int f = ROWS_IN_M ? 4 * (ln >> 4) + s : (ln & 15);
        img[x] = r < L ? tab[(((size_t)r * h + head) * 16 + f) * 3 + ax] : 0.f;
    }
}

// One wave, one (cell piece, head), 16 queries at a time.  The sweeps over a chunk's key tiles are pipelined by hand: what a step
// (axis, key tile) needs from memory - the tile's key rows, its packed rel-pos words - is requested one step ahead, so that only
// the logits of the chunk (4 registers per key tile) live across steps.
template <int LP>
__global__ __launch_bounds__((CmGeo<LP>::WAVES * 64)) void cell_fwd_mfma_kernel(pointops2_cell_plan pl, int h, int L, const float *__restrict__ q,
                                                                                const float *__restrict__ k, const float *__restrict__ v,
                                                                                const float *__restrict__ table_q, const float *__restrict__ table_k,
                                                                                const float *__restrict__ table_v, float *__restrict__ out,
                                                                                float *__restrict__ pbuf, size_t plane) {
    using G = CmGeo<LP>;
    constexpr int NTA = G::NTA, RS = G::RS, TILE = G::TILE, NKT = CM_NKT;
    extern __shared__ float lds[];
    float *img_q = lds, *img_k = lds + G::TAB, *img_v = lds + 2 * G::TAB;
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int head = blockIdx.y, C = h * 16;
    float *qtb = lds + 3 * G::TAB + wave * G::WAVE_FLOATS;  // [16][RS]: QT of one axis, later the weight histogram H (ints)
    float *ktb = qtb + TILE;                                // [16][RS]: KT of one axis for one key tile
    stage_fragments<LP, true>(img_q, table_q, L, h, head);
    stage_fragments<LP, true>(img_k, table_k, L, h, head);
    stage_fragments<LP, false>(img_v, table_v, L, h, head);
    __syncthreads();
    const int nC = share_count(pl, pl.counts[0]);
    float *pb = pbuf + (size_t)head * plane;
    const int slots = gridDim.x * G::WAVES, slot = blockIdx.x * G::WAVES + wave;
    const int hoff = head * 16 + 4 * g;  // this lane's four features of a q / k row (k-slot (s, g) <-> feature 4g + s)
    const f32x4c zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int round = 0; round * slots < nC; round++) {
        const int task = snake_task(round, slot, slots);
        if (task >= nC) continue;
        const CellTask ct = cell_task(pl, share_task(pl, task));
        const unsigned tile_bytes = (unsigned)ct.nq * ct.nk * 4u;
        const rsrc_t rs_rel = make_rsrc(pl.relp + ct.pbase, tile_bytes);
        const rsrc_t rs_key = make_rsrc(pl.cell_keys + ct.kb, (unsigned)ct.nk * 4u);
        const rsrc_t rs_qid = make_rsrc(pl.cell_order + ct.qs, (unsigned)ct.nq * 4u);
        const rsrc_t rs_p = make_rsrc(pb + ct.pbase, tile_bytes);
        const int nch = (ct.nk + 16 * NKT - 1) / (16 * NKT);
        // pieces of more than 16 queries run as consecutive groups of 16
        for (int i0 = 0; i0 < ct.nq; i0 += 16) {
            const bool qok = i0 + n < ct.nq;
            const int qid = (int)bload_u32(rs_qid, (i0 + n) * 4);  // (past the end: 0, never used)
            const float4 qf = qok ? ldg4(q + (size_t)qid * C + hoff) : make_float4(0.f, 0.f, 0.f, 0.f);
            const int row_off = (i0 + n) * ct.nk * 4;  // byte offset of this lane's query row in the tile
            f32x4c acc = zero4;                         // out^T: D[feature 4g + t][query n]
            float run_m = -INFINITY, run_l = 0.f;
            float lg[NKT][4];                           // logits, then softmax weights: [key tile][key 4g + t] of query n

            // ---- sweep 1 of a chunk: logits ----
            auto logits_chunk = [&](int j0, int nkt) {
                int keyid[NKT];
#pragma unroll
                for (int kt = 0; kt < NKT; kt++) keyid[kt] = (int)bload_u32(rs_key, (j0 + 16 * kt + n) * 4);  // (past the end: 0, masked below)
                float4 kf_nx = ldg4(k + (size_t)keyid[0] * C + hoff);
                unsigned w_nx[4];
                bload_words<4>(rs_rel, row_off + (j0 + 4 * g) * 4, w_nx);
#pragma unroll 1
                for (int ax = 0; ax < 3; ax++) {  // (a real loop: unrolled, the kernel is 67 KB of code - more than the instruction cache)
#pragma unroll
                    for (int T = 0; T < NTA; T++) {
                        const float4 a = *reinterpret_cast<const float4 *>(img_q + ((ax * NTA + T) * 256 + lane * 4));
                        const f32x4c d = mfma4(a, qf, zero4);  // D[row 16T + 4g + t][query n]
                        *reinterpret_cast<f32x4c *>(qtb + n * RS + 16 * T + 4 * g) = d;
                    }
#pragma unroll
                    for (int kt = 0; kt < NKT; kt++)
                        if (kt < nkt) {
                            const float4 kf = kf_nx;
                            unsigned w[4];
#pragma unroll
                            for (int t = 0; t < 4; t++) w[t] = w_nx[t];
                            {   // the next step's inputs: the next key tile of this axis, or the first one of the next axis
                                const int kn = kt + 1 < nkt ? kt + 1 : 0;
                                const int key_n = kt + 1 < nkt ? keyid[kt + 1 < NKT ? kt + 1 : 0] : keyid[0];
                                kf_nx = ldg4(k + (size_t)key_n * C + hoff);
                                bload_words<4>(rs_rel, row_off + (j0 + 16 * kn + 4 * g) * 4, w_nx);
                            }
                            if (ax == 0) {
                                const f32x4c s = mfma4(kf, qf, zero4);  // D[key 4g + t][query n]
#pragma unroll
                                for (int t = 0; t < 4; t++) lg[kt][t] = s[t];
                            }
#pragma unroll
                            for (int T = 0; T < NTA; T++) {
                                const float4 a = *reinterpret_cast<const float4 *>(img_k + ((ax * NTA + T) * 256 + lane * 4));
                                const f32x4c d = mfma4(a, kf, zero4);  // D[row 16T + 4g + t][key n]
                                *reinterpret_cast<f32x4c *>(ktb + n * RS + 16 * T + 4 * g) = d;
                            }
                            lds_fence();
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                const int r = (int)((w[t] >> (8 * ax)) & 255u);
                                lg[kt][t] += qtb[n * RS + r] + ktb[(4 * g + t) * RS + r];
                                if (ax == 2) {  // entries that are no pair: past the row's end, unused query slots, flagged candidates
                                    const bool pair = qok && j0 + 16 * kt + 4 * g + t < ct.nk && !(w[t] >> 31);
                                    lg[kt][t] = pair ? lg[kt][t] : -INFINITY;
                                }
                            }
                            __builtin_amdgcn_wave_barrier();  // (the next tile's KT stores stay behind these reads)
                        }
                }
#pragma unroll
                for (int kt = 0; kt < NKT; kt++)
                    if (kt >= nkt) {
#pragma unroll
                        for (int t = 0; t < 4; t++) lg[kt][t] = -INFINITY;
                    }
            };
            // ---- sweep 2 of a chunk: lg holds the softmax weights; out += P V + H Tv ----
            auto values_chunk = [&](int j0, int nkt) {
                unsigned kid_nx[4];
                bload_words<4>(rs_key, (j0 + 4 * g) * 4, kid_nx);  // keys 4g .. 4g + 3 of the tile (past the end: 0, weight 0)
#pragma unroll
                for (int kt = 0; kt < NKT; kt++)
                    if (kt < nkt) {
                        float4 vf;  // A[feature n][k-slot (s, g) <-> key 4g + s]
                        vf.x = v[(size_t)kid_nx[0] * C + head * 16 + n];
                        vf.y = v[(size_t)kid_nx[1] * C + head * 16 + n];
                        vf.z = v[(size_t)kid_nx[2] * C + head * 16 + n];
                        vf.w = v[(size_t)kid_nx[3] * C + head * 16 + n];
                        if (kt + 1 < nkt) bload_words<4>(rs_key, (j0 + 16 * (kt + 1) + 4 * g) * 4, kid_nx);
                        acc = mfma4(vf, make_float4(lg[kt][0], lg[kt][1], lg[kt][2], lg[kt][3]), acc);  // D[feature 4g + t][query n]
                    }
                int *hb = reinterpret_cast<int *>(qtb);
                unsigned w_nx[4];
                bload_words<4>(rs_rel, row_off + (j0 + 4 * g) * 4, w_nx);
#pragma unroll 1
                for (int ax = 0; ax < 3; ax++) {
                    for (int x = lane * 4; x < TILE; x += 256) *reinterpret_cast<int4 *>(hb + x) = make_int4(0, 0, 0, 0);
                    lds_fence();
#pragma unroll
                    for (int kt = 0; kt < NKT; kt++)
                        if (kt < nkt) {
                            unsigned w[4];
#pragma unroll
                            for (int t = 0; t < 4; t++) w[t] = w_nx[t];
                            bload_words<4>(rs_rel, row_off + (j0 + 16 * (kt + 1 < nkt ? kt + 1 : 0) + 4 * g) * 4, w_nx);
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                const float p = lg[kt][t];
                                if (p != 0.f) atomicAdd(hb + n * RS + (int)((w[t] >> (8 * ax)) & 255u), __float2int_rn(p * 1073741824.f));
                            }
                        }
                    lds_fence();
#pragma unroll
                    for (int T = 0; T < NTA; T++) {
                        const int4 hv = *reinterpret_cast<const int4 *>(hb + n * RS + 16 * T + 4 * g);  // H[query n][rows 16T + 4g ..]
                        const float4 a = *reinterpret_cast<const float4 *>(img_v + ((ax * NTA + T) * 256 + lane * 4));
                        const float sc = 1.0f / 1073741824.f;
                        acc = mfma4(a, make_float4((float)hv.x * sc, (float)hv.y * sc, (float)hv.z * sc, (float)hv.w * sc), acc);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            };
            auto store_tile_rows = [&](int j0, int nkt) {  // lg -> pbuf (weights, or parked raw logits)
#pragma unroll
                for (int kt = 0; kt < NKT; kt++)
                    if (kt < nkt) {
                        const int j = j0 + 16 * kt + 4 * g;
                        bstore_floats<4>(rs_p, row_off + j * 4, lg[kt], qok ? min(max(ct.nk - j, 0), 4) : 0);
                    }
            };

            // one pass over the chunks when the whole key list fits the registers (weights made in place), else two: logits parked
            // in pbuf with a running max / sum, then the weights (one call site per sweep keeps the kernel inside the instruction cache)
            const int npass = nch == 1 ? 1 : 2;
#pragma unroll 1
            for (int pass = 0; pass < npass; pass++) {
                const float m_fin = run_m == -INFINITY ? 0.f : run_m;
                const float inv_fin = run_l > 0.f ? 1.0f / run_l : 0.f;
#pragma unroll 1
                for (int ch = 0; ch < nch; ch++) {
                    const int j0 = ch * 16 * NKT, nkt = (min(16 * NKT, ct.nk - j0) + 15) >> 4;
                    if (pass == 0) {
                        logits_chunk(j0, nkt);
                        float mx = -INFINITY;
#pragma unroll
                        for (int kt = 0; kt < NKT; kt++)
#pragma unroll
                            for (int t = 0; t < 4; t++) mx = fmaxf(mx, lg[kt][t]);
                        mx = col_max(mx);
                        const float m_new = fmaxf(run_m, mx);
                        const float m_use = m_new == -INFINITY ? 0.f : m_new;  // (an unused query slot: every weight 0)
                        float sum = 0.f;
                        if (nch == 1) {
#pragma unroll
                            for (int kt = 0; kt < NKT; kt++)
#pragma unroll
                                for (int t = 0; t < 4; t++) {
                                    lg[kt][t] = __expf(lg[kt][t] - m_use);  // exp(-inf) = 0
                                    sum += lg[kt][t];
                                }
                            sum = col_sum(sum);
                            const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
                            for (int kt = 0; kt < NKT; kt++)
#pragma unroll
                                for (int t = 0; t < 4; t++) lg[kt][t] *= inv;
                        } else {
#pragma unroll
                            for (int kt = 0; kt < NKT; kt++)
#pragma unroll
                                for (int t = 0; t < 4; t++) sum += __expf(lg[kt][t] - m_use);
                            sum = col_sum(sum);
                            run_l = (run_m == -INFINITY ? 0.f : run_l * __expf(run_m - m_use)) + sum;
                            run_m = m_new;
                            store_tile_rows(j0, nkt);  // raw logits, parked (-inf where there is no pair)
                            continue;
                        }
                    } else {
#pragma unroll
                        for (int kt = 0; kt < NKT; kt++)
                            if (kt < nkt) bload_floats<4>(rs_p, row_off + (j0 + 16 * kt + 4 * g) * 4, lg[kt]);
#pragma unroll
                        for (int kt = 0; kt < NKT; kt++)
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                const bool in_row = kt < nkt && qok && j0 + 16 * kt + 4 * g + t < ct.nk;  // (parked -inf: exp = 0)
                                lg[kt][t] = in_row ? __expf(lg[kt][t] - m_fin) * inv_fin : 0.f;
                            }
                    }
                    store_tile_rows(j0, nkt);
                    values_chunk(j0, nkt);
                }
            }
            if (qok) *reinterpret_cast<f32x4c *>(out + (size_t)qid * C + hoff) = acc;  // features 4g .. 4g + 3 of query n
        }
    }
}

template <int LP>
static void launch_mfma_fwd(const pointops2_cell_plan *plan, int h, int L, const float *q, const float *k, const float *v, const float *table_q,
                            const float *table_k, const float *table_v, float *out, float *pbuf) {
    using G = CmGeo<LP>;
    const size_t lds = G::lds_bytes();
    allow_big_lds(cell_fwd_mfma_kernel<LP>, lds);
    static const int cf_div = getenv("P2_CF_DIV") ? atoi(getenv("P2_CF_DIV")) : 1;
    const dim3 grid(std::max(1, cell_grid_x(1, plan->n_cells, h, G::WAVES) / cf_div), h);
    hipLaunchKernelGGL((cell_fwd_mfma_kernel<LP>), grid, dim3(G::WAVES * 64), lds, state().stream, *plan, h, L, q, k, v, table_q, table_k, table_v, out,
                       pbuf, (size_t)plan->n_pairs);
}

// fp32 operands, d = 16, L <= 80.  Returns false when the matrix-core forward does not apply (the caller then runs cell_attn.hip's).
bool cell_fwd_mfma_launch(const pointops2_cell_plan *plan, int h, int L, const float *q, const float *k, const float *v, const float *table_q,
                          const float *table_k, const float *table_v, float *out, float *pbuf) {
    // P2_CELL_MFMA: 0 = never, 1 = always (when it applies), unset = where it was measured faster than cell_attn.hip's VALU forward
    // (tools/bench_cell.py, MI355X, the four stages of the S3DIS scene, even / odd pattern, us MFMA : VALU): 292:321 / 372:300,
    // 163:193 / 190:180, 114:152 / 117:120, 103:150 / 92:96.  The matrix-core tiles are 16 queries wide: the shifted pattern of the
    // two large stages cuts the cloud into many cells of ~9 queries (n_pairs / n_keyslots), whose tiles stay half empty.
    static const int mode = getenv("P2_CELL_MFMA") ? atoi(getenv("P2_CELL_MFMA")) : -1;
    if (mode == 0 || L > 80) return false;
    if (mode < 0) {
        const double avg_queries = (double)plan->n_pairs / (double)(plan->n_keyslots > 0 ? plan->n_keyslots : 1);
        if ((long long)plan->n_points * h >= 96000 && avg_queries < 15.0) return false;
    }
    if (L <= 64) launch_mfma_fwd<64>(plan, h, L, q, k, v, table_q, table_k, table_v, out, pbuf);
    else launch_mfma_fwd<80>(plan, h, L, q, k, v, table_q, table_k, table_v, out, pbuf);
    return true;
}

}  // namespace p2
