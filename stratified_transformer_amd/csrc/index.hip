// I3-I5: on-device index build of one Stratified Transformer stage, gfx950.
//
// Reproduces bit-exactly, in the canonical (stable-sort) order, what the reference's model code
// computes with torch ops, [nW,k,k] boolean masks and a 12 M-element sort per block
// (model/stratified_transformer.py:10-65 grid_sample / get_indice_pairs, :186-190 rel-pos index,
// :312-317 CSR) — as a handful of streaming kernels with no masks and no pair sort:
//
//   partition   voxel id per point with the torch_cluster grid_cluster arithmetic (fp32 subtract,
//               IEEE divide, truncate), stable radix sort by voxel id => points bucketed by window
//               with ascending point index inside a bucket, dense window ranks (= torch.unique) by a
//               boundary scan
//   pairs       one wavefront per query: dense keys = its small-window bucket; stratified keys = the
//               FPS-sampled points of its large-window bucket whose fp32 floor-div window coordinate
//               differs from the query's (wave ballot + prefix popcount keeps them ascending);
//               counts -> exclusive scan -> CSR offsets; the fill pass writes index_1, index_0 and the
//               quantised relative position index with torch's own arithmetic (round-half-even,
//               multiply by the fp32 reciprocal of 1e5 as ATen does for a GPU tensor divided by a
//               Python scalar, c10::div_floor_floating for `//`).
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace p2 {

// c10::div_floor_floating (torch `//` on floating tensors), fp32
__device__ __forceinline__ float div_floor(float a, float b) {
    if (b == 0.f) return __fdiv_rn(a, b);
    const float mod = fmodf(a, b);
    float div = __fdiv_rn(__fsub_rn(a, mod), b);
    if ((mod != 0.f) && ((b < 0.f) != (mod < 0.f))) div = __fsub_rn(div, 1.f);
    float floordiv;
    if (div != 0.f) {
        floordiv = floorf(div);
        if (__fsub_rn(div, floordiv) > 0.5f) floordiv = __fadd_rn(floordiv, 1.f);
    } else {
        floordiv = copysignf(0.f, __fdiv_rn(a, b));
    }
    return floordiv;
}

// ---- global bounding box ---------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bbox_kernel(int N, const float *__restrict__ xyz, float *__restrict__ out6) {
    __shared__ float red[6][16];
    const int tid = threadIdx.x;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i0 = tid; i0 < N; i0 += 4 * 1024) {  // four points in flight per thread
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = min(i0 + u * 1024, N - 1);  // (past the end: the last point again)
#pragma unroll
            for (int a = 0; a < 3; a++) v[u][a] = xyz[(size_t)i * 3 + a];
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int a = 0; a < 3; a++) {
                mn[a] = fminf(mn[a], v[u][a]);
                mx[a] = fmaxf(mx[a], v[u][a]);
            }
    }
    for (int a = 0; a < 3; a++) {
        for (int st = 1; st < 64; st <<= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], st, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], st, 64));
        }
        if ((tid & 63) == 0) { red[a][tid >> 6] = mn[a]; red[3 + a][tid >> 6] = mx[a]; }
    }
    __syncthreads();
    if (tid < 6) {
        float v = red[tid][0];
        for (int w = 1; w < 16; w++) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        out6[tid] = v;
    }
}

// ---- partition ---------------------------------------------------------------------------------
__global__ void voxel_key_kernel(int N, int b, const float *__restrict__ xyz, const int *__restrict__ offset,
                                 const float *__restrict__ bbox6, float size, float shift,
                                 unsigned long long *__restrict__ keys, int *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int bid = 0;
    while (bid < b - 1 && i >= offset[bid]) bid++;
    long long id = 0, mult = 1;
    for (int a = 0; a < 3; a++) {
        const float start = bbox6[a];
        const float end = __fadd_rn(bbox6[3 + a], shift);           // max(xyz + shift): fl() is monotone
        const float pos = __fadd_rn(xyz[(size_t)i * 3 + a], shift);
        const long long v = (long long)__fdiv_rn(__fsub_rn(pos, start), size);
        id += v * mult;
        mult *= (long long)__fdiv_rn(__fsub_rn(end, start), size) + 1;
    }
    id += (long long)bid * mult;  // batch is the 4th coordinate: cell size 1, start 0
    keys[i] = (unsigned long long)id;
    vals[i] = i;
}

// The four partitions of a stage (small / small shifted / large / large shifted windows: grid_sample x 4, :277,280,297,300) as ONE
// sort: key = variant | batch | z | y | x with ten bits per coordinate - the reference's voxel id x + y mx + z mx my + batch mx my mz
// orders the voxels of a variant the same way (every coordinate is below its multiplier), so the dense ranks are the same, and the
// key width no longer depends on the bounding box (no host read-back before the sort).  A coordinate that needs more than ten bits
// sets *overflow: the caller reads it with its next read-back and builds the partitions one by one instead.
__global__ void voxel_key4_kernel(int N, int b, const float *__restrict__ xyz, const int *__restrict__ offset, const float *__restrict__ bbox6,
                                  float window, int batch_bits, unsigned long long *__restrict__ keys, int *__restrict__ vals,
                                  int *__restrict__ overflow) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * N) return;
    const int v = t / N, i = t - v * N;
    const float size = v < 2 ? window : __fmul_rn(2.f, window);
    const float shift = v == 1 ? __fmul_rn(0.5f, window) : (v == 3 ? window : 0.f);
    int bid = 0;
    while (bid < b - 1 && i >= offset[bid]) bid++;
    unsigned long long key = (unsigned long long)((v << batch_bits) + bid) << 30;
    bool bad = false;
    for (int a = 0; a < 3; a++) {
        const float pos = __fadd_rn(xyz[(size_t)i * 3 + a], shift);
        const long long c = (long long)__fdiv_rn(__fsub_rn(pos, bbox6[a]), size);
        bad |= c < 0 || c >= 1024;
        key |= (unsigned long long)(c & 1023) << (10 * a);
    }
    if (bad) *overflow = 1;
    keys[t] = key;
    vals[t] = i;
}

__global__ void partition_finish4_kernel(int N, const int *__restrict__ order, const int *__restrict__ flags, const int *__restrict__ rank_incl,
                                         int *__restrict__ cluster, int *__restrict__ starts, int *__restrict__ n_windows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 4 * N) return;
    const int v = t / N, local = t - v * N;
    const int r = rank_incl[t] - rank_incl[v * N];  // (the variant's first element carries a flag: its inclusive rank is the base + 1)
    cluster[(size_t)v * N + order[t]] = r;
    int *st = starts + (size_t)v * (N + 2);
    if (flags[t]) st[r] = local;
    if (local == N - 1) {
        st[r + 1] = N;
        n_windows[v] = r + 1;
    }
}

__global__ void boundary_flag_kernel(int N, const unsigned long long *__restrict__ skeys, int *__restrict__ flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    flags[t] = (t == 0 || skeys[t] != skeys[t - 1]) ? 1 : 0;
}

__global__ void partition_finish_kernel(int N, const int *__restrict__ order, const int *__restrict__ flags,
                                        const int *__restrict__ rank_incl, int *__restrict__ cluster,
                                        int *__restrict__ starts, int *__restrict__ n_windows) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const int r = rank_incl[t] - 1;
    cluster[order[t]] = r;
    if (flags[t]) starts[r] = t;
    if (t == N - 1) {
        starts[r + 1] = N;
        *n_windows = r + 1;
    }
}

// ---- pairs -------------------------------------------------------------------------------------
__global__ void window_coord_kernel(int N, const float *__restrict__ xyz, const float *__restrict__ bbox6, float window,
                                    int shifted, float *__restrict__ wc) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * 3) return;
    const int a = t % 3;
    float v = xyz[t];
    if (shifted) v = __fadd_rn(v, __fmul_rn(0.5f, window));  // xyz + 1/2*window_size   (:32)
    v = __fsub_rn(v, bbox6[a]);                               // - xyz_min
    wc[t] = div_floor(v, window);                             // // window_size
}

__global__ void mark_sampled_kernel(int m, const int *__restrict__ sample_idx, int *__restrict__ sampled) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < m) sampled[sample_idx[t]] = 1;
}
__global__ void sampled_in_order_kernel(int N, const int *__restrict__ l_order, const int *__restrict__ sampled, int *__restrict__ flag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N) flag[t] = sampled[l_order[t]];
}
// compact the sampled points in (large window, point id) order; per-window starts of the compacted list
__global__ void sampled_compact_kernel(int N, const int *__restrict__ l_order, const int *__restrict__ flag,
                                       const int *__restrict__ pos_excl, const int *__restrict__ l_starts,
                                       const int *__restrict__ n_windows, int m_total, int *__restrict__ ls, int *__restrict__ ls_starts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    if (flag[t]) ls[pos_excl[t]] = l_order[t];
    const int nw = *n_windows;
    if (t < nw) ls_starts[t] = pos_excl[l_starts[t]];
    if (t == 0) ls_starts[nw] = m_total;
}

__device__ __forceinline__ bool coord_differs(const float *__restrict__ wc, int i, int j) {
    return wc[(size_t)i * 3] != wc[(size_t)j * 3] || wc[(size_t)i * 3 + 1] != wc[(size_t)j * 3 + 1] ||
           wc[(size_t)i * 3 + 2] != wc[(size_t)j * 3 + 2];
}

__global__ __launch_bounds__(256) void pairs_count_kernel(int N, const int *__restrict__ s_cluster, const int *__restrict__ s_starts,
                                                          const int *__restrict__ l_cluster, const int *__restrict__ ls,
                                                          const int *__restrict__ ls_starts, const float *__restrict__ wc,
                                                          int *__restrict__ total) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int ws = s_cluster[i], wl = l_cluster[i];
    int cnt = s_starts[ws + 1] - s_starts[ws];
    const int c0 = ls_starts[wl], c1 = ls_starts[wl + 1];
    for (int t0 = c0; t0 < c1; t0 += 64) {
        const int t = t0 + lane;
        const bool keep = t < c1 && coord_differs(wc, i, ls[t]);
        cnt += __popcll(__ballot(keep));
    }
    if (lane == 0) total[i] = cnt;
}

__device__ __forceinline__ void write_pair(int i, int j, int at, const float *__restrict__ xyz, float two_w, float quant,
                                           int *__restrict__ index_0, int *__restrict__ index_1, int *__restrict__ rel_idx) {
    index_0[at] = i;
    index_1[at] = j;
    const float inv = __fdiv_rn(1.0f, 100000.0f);
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float r = __fsub_rn(xyz[(size_t)i * 3 + a], xyz[(size_t)j * 3 + a]);  // xyz[index_0] - xyz[index_1]   (:186)
        r = __fmul_rn(rintf(__fmul_rn(r, 100000.0f)), inv);                   // round(. * 1e5) / 1e5          (:187)
        r = __fsub_rn(__fadd_rn(r, two_w), 0.0001f);                          // + 2*window_size - 0.0001      (:188)
        rel_idx[(size_t)at * 3 + a] = (int)div_floor(r, quant);               // // quant_size, .int()
    }
}

__global__ __launch_bounds__(256) void pairs_fill_kernel(int N, const float *__restrict__ xyz, float two_w, float quant,
                                                         const int *__restrict__ s_cluster, const int *__restrict__ s_order,
                                                         const int *__restrict__ s_starts, const int *__restrict__ l_cluster,
                                                         const int *__restrict__ ls, const int *__restrict__ ls_starts,
                                                         const float *__restrict__ wc, const int *__restrict__ offsets,
                                                         int *__restrict__ index_0, int *__restrict__ index_1, int *__restrict__ rel_idx) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int ws = s_cluster[i], wl = l_cluster[i];
    int at = offsets[i];
    const int d0 = s_starts[ws], d1 = s_starts[ws + 1];
    for (int t = d0 + lane; t < d1; t += 64) write_pair(i, s_order[t], at + (t - d0), xyz, two_w, quant, index_0, index_1, rel_idx);
    at += d1 - d0;
    const int c0 = ls_starts[wl], c1 = ls_starts[wl + 1];
    for (int t0 = c0; t0 < c1; t0 += 64) {
        const int t = t0 + lane;
        const int j = t < c1 ? ls[t] : 0;
        const bool keep = t < c1 && coord_differs(wc, i, j);
        const unsigned long long mask = __ballot(keep);
        if (keep) write_pair(i, j, at + __popcll(mask & ((1ull << lane) - 1)), xyz, two_w, quant, index_0, index_1, rel_idx);
        at += __popcll(mask);
    }
}


// ---- cells: the window-centric view of a pair list (cell_attn.hip) -------------------
// All queries with the same (small window, large window) share their candidate key list: the small window's
// bucket (dense keys) followed by the sampled points of the large window's bucket (stratified candidates);
// a candidate is a key of query i iff its fp32 floor-div window coordinate differs from i's - the very
// predicate of pairs_fill_kernel, evaluated per (query, candidate) and stored as a flag.  A cell is therefore a
// dense n_q x n_k tile: the attention kernels load a cell's key rows once instead of once per query.
__global__ void cell_key_kernel(int N, int lbits, const int *__restrict__ s_cluster, const int *__restrict__ l_cluster,
                                unsigned long long *__restrict__ keys, int *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    keys[i] = ((unsigned long long)(unsigned)s_cluster[i] << lbits) | (unsigned)l_cluster[i];
    vals[i] = i;
}
// Cells are cut into pieces of at most `cap` queries (same key list, a tile of its own): a piece is the unit of work of
// one wave, and the coarse stages have few intersections but many heads' worth of CUs to fill.
__global__ void cell_run_start_kernel(int N, const int *__restrict__ flags, const int *__restrict__ rank_incl, int *__restrict__ run_start) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N && flags[t]) run_start[rank_incl[t] - 1] = t;
}
__global__ void cell_split_flag_kernel(int N, int cap, const int *__restrict__ rank_incl, const int *__restrict__ run_start, int *__restrict__ flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N && (t - run_start[rank_incl[t] - 1]) % cap == 0) flags[t] = 1;
}
// parents = the uncut cells: first piece of every parent (pieces of a parent are consecutive cell ids with the same key
// list, so their tiles are one contiguous [sum n_q, n_k] tile: the key-side table gradient walks parents)
__global__ void cell_parent_kernel(int N, const int *__restrict__ split_flags, const int *__restrict__ split_rank,
                                   const int *__restrict__ parent_rank, const int *__restrict__ run_start, int *__restrict__ parent_first,
                                   int *__restrict__ counts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const int parent = parent_rank[t] - 1;
    if (split_flags[t] && run_start[parent] == t) parent_first[parent] = split_rank[t] - 1;
    if (t == N - 1) {
        parent_first[parent + 1] = split_rank[t];
        counts[4] = parent + 1;
    }
}
// cell id of every sorted position; per cell: first sorted position, descriptor {dense start, dense count,
// candidate start, candidate count}, key count and tile size (for the scans)
__global__ void cell_describe_kernel(int N, const int *__restrict__ order, const int *__restrict__ flags, const int *__restrict__ rank_incl,
                                     const int *__restrict__ s_cluster, const int *__restrict__ s_starts,
                                     const int *__restrict__ l_cluster, const int *__restrict__ ls_starts, int *__restrict__ qcell,
                                     int *__restrict__ cell_qstart, int *__restrict__ cell_desc, int *__restrict__ nk_of,
                                     int *__restrict__ counts) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const int cid = rank_incl[t] - 1;
    qcell[t] = cid;
    if (flags[t]) {
        const int i = order[t];
        const int ws = s_cluster[i], wl = l_cluster[i];
        const int d0 = s_starts[ws], nd = s_starts[ws + 1] - d0;
        const int c0 = ls_starts[wl], ns = ls_starts[wl + 1] - c0;
        cell_qstart[cid] = t;
        cell_desc[cid * 4 + 0] = d0;
        cell_desc[cid * 4 + 1] = nd;
        cell_desc[cid * 4 + 2] = c0;
        cell_desc[cid * 4 + 3] = ns;
        nk_of[cid] = nd + ns;
        atomicMax(&counts[3], nd + ns);
    }
    if (t == N - 1) {
        cell_qstart[cid + 1] = N;
        counts[0] = cid + 1;
    }
}
// tile sizes n_q * n_k (0 past the last cell) and the work-order sort keys (largest tile first)
__global__ void cell_tile_kernel(int N, const int *__restrict__ counts, const int *__restrict__ cell_qstart, int *__restrict__ nk_of,
                                 int *__restrict__ tile_of, unsigned *__restrict__ work_key, int *__restrict__ ids) {
    const int cid = blockIdx.x * blockDim.x + threadIdx.x;
    if (cid > N) return;
    const int nC = counts[0];
    int nk = 0, tile = 0;
    if (cid < nC) {
        nk = nk_of[cid];
        tile = (cell_qstart[cid + 1] - cell_qstart[cid]) * nk;
    }
    nk_of[cid] = nk;
    tile_of[cid] = tile;
    if (cid < N) {
        work_key[cid] = 0xffffffffu - (unsigned)tile;
        ids[cid] = cid;
    }
}
__global__ void cell_totals_kernel(int N, const int *__restrict__ cell_kbase, const int *__restrict__ cell_pbase, int *__restrict__ counts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        counts[1] = cell_pbase[N];  // pair slots (tile entries) of all cells
        counts[2] = cell_kbase[N];  // key slots of all cells
    }
}
// one wave per sorted query position: its row of the cell's tile = packed rel-pos index (r0 | r1 << 8 | r2 << 16) of
// every candidate key, bit 31 set where the candidate is not a key of this query; the wave of a cell's first query
// also writes the cell's key list
__global__ __launch_bounds__(256) void cell_fill_kernel(int N, const float *__restrict__ xyz, float two_w, float quant, int L,
                                                        const int *__restrict__ s_order, const int *__restrict__ ls,
                                                        const float *__restrict__ wc, const int *__restrict__ order,
                                                        const int *__restrict__ qcell, const int *__restrict__ cell_qstart,
                                                        const int *__restrict__ cell_desc, const int *__restrict__ cell_kbase,
                                                        const int *__restrict__ cell_pbase, int *__restrict__ cell_keys,
                                                        int *__restrict__ kcell, unsigned *__restrict__ relp) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + wave;
    if (t >= N) return;
    const int cid = qcell[t], i = order[t];
    const int d0 = cell_desc[cid * 4], nd = cell_desc[cid * 4 + 1], c0 = cell_desc[cid * 4 + 2], ns = cell_desc[cid * 4 + 3];
    const int nk = nd + ns, il = t - cell_qstart[cid], kb = cell_kbase[cid];
    unsigned *row = relp + (size_t)cell_pbase[cid] + (size_t)il * nk;
    const float inv = __fdiv_rn(1.0f, 100000.0f);
    const float xi[3] = {xyz[(size_t)i * 3], xyz[(size_t)i * 3 + 1], xyz[(size_t)i * 3 + 2]};
    for (int jl = lane; jl < nk; jl += 64) {
        const int j = jl < nd ? s_order[d0 + jl] : ls[c0 + jl - nd];
        unsigned w = (jl >= nd && !coord_differs(wc, i, j)) ? 0x80000000u : 0u;
#pragma unroll
        for (int a = 0; a < 3; a++) {  // the arithmetic of write_pair
            float r = __fsub_rn(xi[a], xyz[(size_t)j * 3 + a]);
            r = __fmul_rn(rintf(__fmul_rn(r, 100000.0f)), inv);
            r = __fsub_rn(__fadd_rn(r, two_w), 0.0001f);
            const int b = min(max((int)div_floor(r, quant), 0), L - 1);  // the model asserts 0 <= . < L (:189-190)
            w |= (unsigned)b << (8 * a);
        }
        row[jl] = w;
        if (il == 0) {
            cell_keys[kb + jl] = j;
            kcell[kb + jl] = cid;
        }
    }
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static size_t sort64_bytes(int N) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (const int *)nullptr, (int *)nullptr, N, 0, 64, (hipStream_t) nullptr);
    return bytes;
}
static size_t sort32_bytes(int N) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned *)nullptr, (unsigned *)nullptr, (const int *)nullptr, (int *)nullptr, N, 0, 32,
                                             (hipStream_t) nullptr);
    return bytes;
}
// key of a row in the window order: its first partner (the pair lists hold a window's own points first, ascending: the first partner
// of every query of a window is the window's lowest point id); rows without partners go last
__global__ void row_order_key_kernel(int N, int M, const int *__restrict__ offs, const int *__restrict__ idx1, unsigned *__restrict__ keys,
                                     int *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int s = offs[i], e = offs[i + 1];
    keys[i] = (s >= 0 && s < e && s < M) ? (unsigned)idx1[s] : 0xffffffffu;
    vals[i] = i;
}
static size_t scan_bytes(int N) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::InclusiveSum(nullptr, bytes, (const int *)nullptr, (int *)nullptr, N, (hipStream_t) nullptr);
    size_t b2 = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b2, (const int *)nullptr, (int *)nullptr, N, (hipStream_t) nullptr);
    return bytes > b2 ? bytes : b2;
}

}  // namespace p2

using namespace p2;

extern "C" {

void pointops2_bbox_launcher(int N, const float *xyz, float *out6) {
    if (N <= 0) return;
    hipLaunchKernelGGL(bbox_kernel, dim3(1), dim3(1024), 0, state().stream, N, xyz, out6);
    check_launch();
}

size_t pointops2_index_workspace_bytes(int N) {
    if (N <= 0) return 0;
    return 2 * al((size_t)N * 8) + 4 * al(((size_t)N + 1) * 4) + al(sort64_bytes(N)) + al(scan_bytes(N + 1));
}

// key_bits: number of significant bits of the voxel ids (the caller knows the bounding box); 0 = sort all 64
void pointops2_window_partition_launcher(int N, int b, const float *xyz, const int *offset, const float *bbox6, float size,
                                         float shift, int key_bits, int *cluster, int *order, int *starts, int *n_windows, void *ws,
                                         size_t ws_bytes) {
    if (N <= 0) return;
    if (ws_bytes < pointops2_index_workspace_bytes(N)) { set_error("pointops2_window_partition: workspace too small"); return; }
    hipStream_t st = state().stream;
    char *p = reinterpret_cast<char *>(ws);
    unsigned long long *keys_in = (unsigned long long *)p; p += al((size_t)N * 8);
    unsigned long long *keys_out = (unsigned long long *)p; p += al((size_t)N * 8);
    int *vals_in = (int *)p; p += al(((size_t)N + 1) * 4);
    int *flags = (int *)p; p += al(((size_t)N + 1) * 4);
    int *rank = (int *)p; p += al(((size_t)N + 1) * 4);
    p += al(((size_t)N + 1) * 4);
    void *tmp = p;
    size_t tmp_bytes = ws_bytes - (size_t)(p - reinterpret_cast<char *>(ws));
    const int g = div_up(N, 256);
    hipLaunchKernelGGL(voxel_key_kernel, dim3(g), dim3(256), 0, st, N, b, xyz, offset, bbox6, size, shift, keys_in, vals_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (const unsigned long long *)keys_in, keys_out, (const int *)vals_in, order, N,
                                                      0, (key_bits > 0 && key_bits < 64) ? key_bits : 64, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(boundary_flag_kernel, dim3(g), dim3(256), 0, st, N, keys_out, flags);
    e = hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, (const int *)flags, rank, N, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(partition_finish_kernel, dim3(g), dim3(256), 0, st, N, order, flags, rank, cluster, starts, n_windows);
    check_launch();
}

size_t pointops2_row_order_workspace_bytes(int N) {
    if (N <= 0) return 0;
    return 2 * al((size_t)N * 4) + al((size_t)N * 4) + al(sort32_bytes(N));
}

// order [N]: the rows of the CSR pair list (offsets [N+1], index1 [M]) sorted by their first partner, ties by row id - rows of one
// window become neighbours (common.h, rows_in_order)
void pointops2_row_order_launcher(int N, int M, const int *offsets, const int *index1, int *order, void *ws, size_t ws_bytes) {
    if (N <= 0) return;
    if (ws_bytes < pointops2_row_order_workspace_bytes(N)) { set_error("pointops2_row_order: workspace too small"); return; }
    hipStream_t st = state().stream;
    char *p = reinterpret_cast<char *>(ws);
    unsigned *keys_in = (unsigned *)p; p += al((size_t)N * 4);
    unsigned *keys_out = (unsigned *)p; p += al((size_t)N * 4);
    int *vals_in = (int *)p; p += al((size_t)N * 4);
    size_t tmp_bytes = ws_bytes - (size_t)(p - reinterpret_cast<char *>(ws));
    hipLaunchKernelGGL(row_order_key_kernel, dim3(div_up(N, 256)), dim3(256), 0, st, N, M, offsets, index1, keys_in, vals_in);
    // (all 32 bits: rows without partners carry 0xffffffff)
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(p, tmp_bytes, (const unsigned *)keys_in, keys_out, (const int *)vals_in, order, N, 0, 32, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    check_launch();
}

size_t pointops2_partitions4_workspace_bytes(int N) {
    if (N <= 0) return 0;
    const size_t n4 = 4 * (size_t)N;
    return 2 * al(n4 * 8) + 3 * al((n4 + 1) * 4) + al(sort64_bytes((int)n4)) + al(scan_bytes((int)n4 + 1));
}

// cluster / order [4][N], starts [4][N+2], n_windows [4]: variant 0 small, 1 small shifted (window/2), 2 large (2 window), 3 large shifted
// (window); *overflow = 1 when a voxel coordinate does not fit the fixed key (the outputs are then in range but meaningless)
void pointops2_window_partitions4_launcher(int N, int b, const float *xyz, const int *offset, const float *bbox6, float window, int *cluster,
                                           int *order, int *starts, int *n_windows, int *overflow, void *ws, size_t ws_bytes) {
    if (N <= 0) return;
    if ((size_t)N * 4 >= (size_t)1 << 30) { set_error("pointops2_window_partitions4: too many points"); return; }
    if (ws_bytes < pointops2_partitions4_workspace_bytes(N)) { set_error("pointops2_window_partitions4: workspace too small"); return; }
    hipStream_t st = state().stream;
    const int n4 = 4 * N;
    char *p = reinterpret_cast<char *>(ws);
    unsigned long long *keys_in = (unsigned long long *)p; p += al((size_t)n4 * 8);
    unsigned long long *keys_out = (unsigned long long *)p; p += al((size_t)n4 * 8);
    int *vals_in = (int *)p; p += al(((size_t)n4 + 1) * 4);
    int *flags = (int *)p; p += al(((size_t)n4 + 1) * 4);
    int *rank = (int *)p; p += al(((size_t)n4 + 1) * 4);
    void *tmp = p;
    size_t tmp_bytes = ws_bytes - (size_t)(p - reinterpret_cast<char *>(ws));
    int batch_bits = 0;
    while ((1 << batch_bits) < b) batch_bits++;
    const int g = div_up(n4, 256);
    (void)hipMemsetAsync(overflow, 0, sizeof(int), st);
    hipLaunchKernelGGL(voxel_key4_kernel, dim3(g), dim3(256), 0, st, N, b, xyz, offset, bbox6, window, batch_bits, keys_in, vals_in, overflow);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (const unsigned long long *)keys_in, keys_out, (const int *)vals_in, order, n4,
                                                      0, 32 + batch_bits, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(boundary_flag_kernel, dim3(g), dim3(256), 0, st, n4, keys_out, flags);
    e = hipcub::DeviceScan::InclusiveSum(tmp, tmp_bytes, (const int *)flags, rank, n4, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(partition_finish4_kernel, dim3(g), dim3(256), 0, st, N, order, flags, rank, cluster, starts, n_windows);
    check_launch();
}

void pointops2_window_coord_launcher(int N, const float *xyz, const float *bbox6, float window, int shifted, float *wc) {
    if (N <= 0) return;
    hipLaunchKernelGGL(window_coord_kernel, dim3(div_up(N * 3, 256)), dim3(256), 0, state().stream, N, xyz, bbox6, window, shifted, wc);
    check_launch();
}

// sampled points of one large-window partition, compacted in (window, point id) order
void pointops2_sampled_buckets_launcher(int N, int m, const int *sample_idx, const int *l_order, const int *l_starts,
                                        const int *l_n_windows, int *sampled /*[N] zeroed by the caller*/, int *ls /*[m]*/,
                                        int *ls_starts /*[N+1]*/, void *ws, size_t ws_bytes) {
    if (N <= 0) return;
    if (ws_bytes < pointops2_index_workspace_bytes(N)) { set_error("pointops2_sampled_buckets: workspace too small"); return; }
    hipStream_t st = state().stream;
    char *p = reinterpret_cast<char *>(ws);
    int *flag = (int *)p; p += al(((size_t)N + 1) * 4);
    int *pos = (int *)p; p += al(((size_t)N + 1) * 4);
    void *tmp = p;
    size_t tmp_bytes = ws_bytes - (size_t)(p - reinterpret_cast<char *>(ws));
    const int g = div_up(N, 256);
    if (m > 0) hipLaunchKernelGGL(mark_sampled_kernel, dim3(div_up(m, 256)), dim3(256), 0, st, m, sample_idx, sampled);
    hipLaunchKernelGGL(sampled_in_order_kernel, dim3(g), dim3(256), 0, st, N, l_order, sampled, flag);
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, (const int *)flag, pos, N, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(sampled_compact_kernel, dim3(g), dim3(256), 0, st, N, l_order, flag, pos, l_starts, l_n_windows, m, ls, ls_starts);
    check_launch();
}

// pass 1: offsets[N+1] (exclusive scan of the per-query key counts; offsets[N] = M)
void pointops2_pairs_count_launcher(int N, const int *s_cluster, const int *s_starts, const int *l_cluster, const int *ls,
                                    const int *ls_starts, const float *wc, int *offsets, void *ws, size_t ws_bytes) {
    if (N <= 0) return;
    if (ws_bytes < pointops2_index_workspace_bytes(N)) { set_error("pointops2_pairs_count: workspace too small"); return; }
    hipStream_t st = state().stream;
    char *p = reinterpret_cast<char *>(ws);
    int *total = (int *)p; p += al(((size_t)N + 1) * 4);
    void *tmp = p;
    size_t tmp_bytes = ws_bytes - (size_t)(p - reinterpret_cast<char *>(ws));
    (void)hipMemsetAsync(total + N, 0, sizeof(int), st);
    hipLaunchKernelGGL(pairs_count_kernel, dim3(div_up(N, 4)), dim3(256), 0, st, N, s_cluster, s_starts, l_cluster, ls, ls_starts, wc, total);
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, (const int *)total, offsets, N + 1, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    check_launch();
}

// pass 2: index_0 / index_1 [M], rel_idx [M,3]
void pointops2_pairs_fill_launcher(int N, const float *xyz, float window, float quant, const int *s_cluster, const int *s_order,
                                   const int *s_starts, const int *l_cluster, const int *ls, const int *ls_starts, const float *wc,
                                   const int *offsets, int *index_0, int *index_1, int *rel_idx) {
    if (N <= 0) return;
    // 2 * self.window_size is a Python float product rounded to fp32 when added to the tensor (:188)
    const float two_w = (float)(2.0 * (double)window);
    hipLaunchKernelGGL(pairs_fill_kernel, dim3(div_up(N, 4)), dim3(256), 0, state().stream, N, xyz, two_w, quant, s_cluster, s_order, s_starts,
                       l_cluster, ls, ls_starts, wc, offsets, index_0, index_1, rel_idx);
    check_launch();
}


// ---- cells (see cell_key_kernel) ------------------------------------------------------------------------------
size_t pointops2_cell_plan_workspace_bytes(int N) {
    if (N <= 0) return 0;
    return 2 * al((size_t)N * 8) + 6 * al(((size_t)N + 2) * 4) + al(sort64_bytes(N)) + al(scan_bytes(N + 2));
}

// pass 1 (no host sync needed before it): cells of one block pattern from its small / large partitions and the
// bucketed samples; max_queries > 0 cuts cells into pieces of at most that many queries.  All outputs caller-allocated: cell_order, qcell, cell_perm [N]; cell_desc [4N]; cell_qstart,
// cell_kbase, cell_pbase, parent_first [N+2]; counts [8] = {cells, tile entries P, key slots K, largest key count, parents}.
// The workspace of one pattern's plan: the first pass leaves flags / rank / the uncut ranks in it for the second
struct CellWs {
    unsigned long long *keys_in, *keys_out;
    int *vals_in, *flags, *rank, *nk_of, *tile_of;
    unsigned *work_key;
    void *tmp;
    size_t tmp_bytes;
};
static CellWs cell_ws(int N, void *ws, size_t ws_bytes) {
    CellWs w;
    char *p = reinterpret_cast<char *>(ws);
    w.keys_in = (unsigned long long *)p; p += al((size_t)N * 8);
    w.keys_out = (unsigned long long *)p; p += al((size_t)N * 8);
    w.vals_in = (int *)p; p += al(((size_t)N + 2) * 4);
    w.flags = (int *)p; p += al(((size_t)N + 2) * 4);
    w.rank = (int *)p; p += al(((size_t)N + 2) * 4);
    w.nk_of = (int *)p; p += al(((size_t)N + 2) * 4);
    w.tile_of = (int *)p; p += al(((size_t)N + 2) * 4);
    w.work_key = (unsigned *)p; p += al(((size_t)N + 2) * 4);
    w.tmp = p;
    w.tmp_bytes = ws_bytes - (size_t)(p - reinterpret_cast<char *>(ws));
    return w;
}

// pass 1a: what needs the two partitions only - the cells (points sorted by (small, large) window), their cut into pieces of at most
// max_queries queries, the parents.  A caller runs it BESIDE the stage's sampler; `ws` must stay untouched until pass 1b has run.
void pointops2_cell_plan_prepare_launcher(int N, int max_queries, const int *s_cluster, const int *l_cluster, int *cell_order,
                                          int *parent_first, int *counts, void *ws, size_t ws_bytes) {
    if (N <= 0) return;
    if (ws_bytes < pointops2_cell_plan_workspace_bytes(N)) { set_error("pointops2_cell_plan_prepare: workspace too small"); return; }
    hipStream_t st = state().stream;
    CellWs w = cell_ws(N, ws, ws_bytes);
    const int g = div_up(N, 256);
    int lbits = 1;
    while ((1ll << lbits) <= (long long)N) lbits++;  // window ids are < N
    (void)hipMemsetAsync(counts, 0, 8 * sizeof(int), st);
    hipLaunchKernelGGL(cell_key_kernel, dim3(g), dim3(256), 0, st, N, lbits, s_cluster, l_cluster, w.keys_in, w.vals_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(w.tmp, w.tmp_bytes, (const unsigned long long *)w.keys_in, w.keys_out,
                                                      (const int *)w.vals_in, cell_order, N, 0, 2 * lbits, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(boundary_flag_kernel, dim3(g), dim3(256), 0, st, N, w.keys_out, w.flags);
    e = hipcub::DeviceScan::InclusiveSum(w.tmp, w.tmp_bytes, (const int *)w.flags, w.rank, N, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    // (nk_of, tile_of and work_key are still free: first position of every uncut run, uncut rank, cut rank)
    int *prank = w.tile_of, *srank = reinterpret_cast<int *>(w.work_key);
    hipLaunchKernelGGL(cell_run_start_kernel, dim3(g), dim3(256), 0, st, N, w.flags, w.rank, w.nk_of);
    (void)hipMemcpyAsync(prank, w.rank, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, st);
    if (max_queries > 0) {
        hipLaunchKernelGGL(cell_split_flag_kernel, dim3(g), dim3(256), 0, st, N, max_queries, w.rank, w.nk_of, w.flags);
        e = hipcub::DeviceScan::InclusiveSum(w.tmp, w.tmp_bytes, (const int *)w.flags, w.rank, N, st);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    }
    (void)hipMemcpyAsync(srank, w.rank, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, st);
    hipLaunchKernelGGL(cell_parent_kernel, dim3(g), dim3(256), 0, st, N, w.flags, srank, prank, w.nk_of, parent_first, counts);
    check_launch();
}

// pass 1b: what needs the sampled points too (ls_starts) - key counts, tile sizes, their scans, the work order, the totals
void pointops2_cell_plan_sizes_launcher(int N, const int *s_cluster, const int *s_starts, const int *l_cluster, const int *ls_starts,
                                        const int *cell_order, int *qcell, int *cell_desc, int *cell_qstart, int *cell_kbase,
                                        int *cell_pbase, int *cell_perm, int *counts, void *ws, size_t ws_bytes) {
    if (N <= 0) return;
    if (ws_bytes < pointops2_cell_plan_workspace_bytes(N)) { set_error("pointops2_cell_plan_sizes: workspace too small"); return; }
    hipStream_t st = state().stream;
    CellWs w = cell_ws(N, ws, ws_bytes);
    const int g = div_up(N, 256);
    hipLaunchKernelGGL(cell_describe_kernel, dim3(g), dim3(256), 0, st, N, cell_order, w.flags, w.rank, s_cluster, s_starts, l_cluster, ls_starts,
                       qcell, cell_qstart, cell_desc, w.nk_of, counts);
    // (vals_in is free again: the ids of the work-order sort)
    hipLaunchKernelGGL(cell_tile_kernel, dim3(div_up(N + 1, 256)), dim3(256), 0, st, N, counts, cell_qstart, w.nk_of, w.tile_of, w.work_key, w.vals_in);
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(w.tmp, w.tmp_bytes, (const int *)w.nk_of, cell_kbase, N + 1, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    e = hipcub::DeviceScan::ExclusiveSum(w.tmp, w.tmp_bytes, (const int *)w.tile_of, cell_pbase, N + 1, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    e = hipcub::DeviceRadixSort::SortPairs(w.tmp, w.tmp_bytes, (const unsigned *)w.work_key, (unsigned *)w.keys_out, (const int *)w.vals_in,
                                           cell_perm, N, 0, 32, st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return; }
    hipLaunchKernelGGL(cell_totals_kernel, dim3(1), dim3(64), 0, st, N, cell_kbase, cell_pbase, counts);
    check_launch();
}

// pass 1 in one call (1a + 1b)
void pointops2_cell_plan_count_launcher(int N, int max_queries, const int *s_cluster, const int *s_starts, const int *l_cluster,
                                        const int *ls_starts, int *cell_order, int *qcell, int *cell_desc, int *cell_qstart,
                                        int *cell_kbase, int *cell_pbase, int *cell_perm, int *parent_first, int *counts, void *ws,
                                        size_t ws_bytes) {
    pointops2_cell_plan_prepare_launcher(N, max_queries, s_cluster, l_cluster, cell_order, parent_first, counts, ws, ws_bytes);
    if (state().error != nullptr) return;
    pointops2_cell_plan_sizes_launcher(N, s_cluster, s_starts, l_cluster, ls_starts, cell_order, qcell, cell_desc, cell_qstart, cell_kbase,
                                       cell_pbase, cell_perm, counts, ws, ws_bytes);
}

// pass 2 (the caller has read counts and allocated cell_keys / kcell [K] and relp [P]): key lists and packed rel-pos tiles
void pointops2_cell_plan_fill_launcher(int N, const float *xyz, float window, float quant, int L, const int *s_order, const int *ls,
                                       const float *wc, const int *cell_order, const int *qcell, const int *cell_qstart,
                                       const int *cell_desc, const int *cell_kbase, const int *cell_pbase, int *cell_keys, int *kcell,
                                       unsigned *relp) {
    if (N <= 0) return;
    if (L < 1 || L > 255) { set_error("pointops2_cell_plan_fill: table rows L must be in 1..255 (packed rel-pos index)"); return; }
    const float two_w = (float)(2.0 * (double)window);
    hipLaunchKernelGGL(cell_fill_kernel, dim3(div_up(N, 4)), dim3(256), 0, state().stream, N, xyz, two_w, quant, L, s_order, ls, wc, cell_order,
                       qcell, cell_qstart, cell_desc, cell_kbase, cell_pbase, cell_keys, kcell, relp);
    check_launch();
}

}  // extern "C"
