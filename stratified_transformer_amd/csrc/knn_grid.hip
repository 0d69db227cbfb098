// I2 (fast path): exact kNN through a uniform grid, gfx950.
//
// The reference's kNN is a full scan per query (knnquery_cuda_kernel.cu:92-102): m*n distance
// evaluations (2.5e9 at the first TransitionDown of a 100k-point scene).  Its RESULT, however, is
// fully determined by the k+1 smallest distances whenever those are pairwise distinct: the heap ends
// up holding exactly the k smallest and heap_sort emits them ascending (proof sketch in DESIGN.md).
// Only exact distance ties make the outcome depend on the heap's history.  So:
//
//   1. candidates are binned into a uniform grid (cell ~ 2*cbrt(V/n), at most 2^18 cells per batch
//      element), sorted by cell with a stable radix sort, and copied into cell order as
//      (x, y, z, original index) records;
//   2. one thread per query walks the cells in growing cubic shells around its own cell and keeps the
//      k+1 best (d2, index) in an LDS column; it stops when the (k+1)-th best distance is smaller than
//      the distance to the nearest unvisited cell (with a safety margin for fp32 cell assignment);
//      distances use the same fma chain as the reference arithmetic;
//   3. a query whose k+1 best contain two equal distances is put on a replay list and recomputed by
//      the literal heap procedure of knn.hip (bit-exact ties); everything else is final.
//
// Needs scratch memory (pointops2_set_workspace) and the candidate / batch counts
// (pointops2_set_point_count, pointops2_set_batch_count); without them knn.hip's full scan runs.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace p2 {

constexpr int KNN_CELL_CAP = 1 << 18;  // cells per batch element

struct KnnPlan {
    float lo[3];
    float c, inv_c;
    int dims[3];
    int ncell;
};

__global__ void knn_plan_kernel(int b, int n, const float *__restrict__ bbox, KnnPlan *plan) {
    if (threadIdx.x || blockIdx.x) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < b; i++)
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], bbox[i * 6 + a]);
            hi[a] = fmaxf(hi[a], bbox[i * 6 + 3 + a]);
        }
    float ext[3];
    double vol = 1.0;
    for (int a = 0; a < 3; a++) {
        ext[a] = fmaxf(hi[a] - lo[a], 1e-6f);
        vol *= ext[a];
    }
    const float per_batch = fmaxf((float)n / (float)b, 1.f);
    float c = 2.f * cbrtf((float)vol / per_batch);
    c = fmaxf(c, 1e-6f);
    for (;;) {  // respect the cell cap
        long long tot = 1;
        for (int a = 0; a < 3; a++) tot *= (long long)(ext[a] / c) + 1;
        if (tot <= KNN_CELL_CAP) break;
        c *= 1.26f;
    }
    plan->c = c;
    plan->inv_c = 1.f / c;
    int ncell = 1;
    for (int a = 0; a < 3; a++) {
        plan->lo[a] = lo[a];
        plan->dims[a] = (int)(ext[a] / c) + 1;
        ncell *= plan->dims[a];
    }
    plan->ncell = ncell;
}

__device__ __forceinline__ int cell_coord(float x, float lo, float inv_c, int dim) {
    return min(max((int)((x - lo) * inv_c), 0), dim - 1);
}

__global__ void knn_cell_kernel(int n, int b, const float *__restrict__ xyz, const int *__restrict__ offset,
                                const KnnPlan *__restrict__ plan, unsigned *__restrict__ keys, int *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bid = 0;
    while (bid < b - 1 && i >= offset[bid]) bid++;
    const KnnPlan p = *plan;
    const int cx = cell_coord(xyz[(size_t)i * 3 + 0], p.lo[0], p.inv_c, p.dims[0]);
    const int cy = cell_coord(xyz[(size_t)i * 3 + 1], p.lo[1], p.inv_c, p.dims[1]);
    const int cz = cell_coord(xyz[(size_t)i * 3 + 2], p.lo[2], p.inv_c, p.dims[2]);
    keys[i] = (unsigned)(bid * p.ncell + (cz * p.dims[1] + cy) * p.dims[0] + cx);
    vals[i] = i;
}

// after the stable sort by cell: cell_start for every cell of every batch element, records in cell order
__global__ void knn_finish_kernel(int n, int b, const unsigned *__restrict__ skeys, const int *__restrict__ svals,
                                  const float *__restrict__ xyz, const KnnPlan *__restrict__ plan,
                                  int *__restrict__ cell_start, float4 *__restrict__ rec) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int total = plan->ncell * b;
    const int kcur = (int)skeys[t];
    const int kprev = t == 0 ? -1 : (int)skeys[t - 1];
    for (int kk = kprev + 1; kk <= kcur; kk++) cell_start[kk] = t;
    if (t == n - 1)
        for (int kk = kcur + 1; kk <= total; kk++) cell_start[kk] = n;
    const int o = svals[t];
    rec[t] = make_float4(xyz[(size_t)o * 3 + 0], xyz[(size_t)o * 3 + 1], xyz[(size_t)o * 3 + 2], __int_as_float(o));
}

template <int BS>
__global__ __launch_bounds__(BS) void knn_grid_kernel(int m, int k, const float *__restrict__ new_xyz,
                                                      const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                      const KnnPlan *__restrict__ plan, const int *__restrict__ cell_start,
                                                      const float4 *__restrict__ rec, int *__restrict__ idx,
                                                      float *__restrict__ dist2, int *__restrict__ replay,
                                                      int *__restrict__ replay_count, const int *__restrict__ qperm) {
    extern __shared__ float smem[];
    float *ld = smem;                                           // [k+1][BS] ascending (d2, index)
    int *li = reinterpret_cast<int *>(smem + (size_t)(k + 1) * BS);
    const int tid = threadIdx.x;
    if (blockIdx.x * BS + tid >= m) return;
    // queries are taken in CELL order (qperm: the queries sorted by their own grid cell): the 64 queries of a wave
    // then walk the same few cells - the same cell_start entries and candidate records, similar trip counts -
    // instead of 64 unrelated neighbourhoods (FPS order scatters consecutive samples over the whole cloud)
    const int pt = qperm[blockIdx.x * BS + tid];
    int bt = 0;
    while (!(pt < new_offset[bt])) bt++;
    const int start = bt == 0 ? 0 : offset[bt - 1];
    const KnnPlan p = *plan;
    const float qx = new_xyz[(size_t)pt * 3 + 0], qy = new_xyz[(size_t)pt * 3 + 1], qz = new_xyz[(size_t)pt * 3 + 2];
    const int cx = cell_coord(qx, p.lo[0], p.inv_c, p.dims[0]);
    const int cy = cell_coord(qy, p.lo[1], p.inv_c, p.dims[1]);
    const int cz = cell_coord(qz, p.lo[2], p.inv_c, p.dims[2]);
    const int base = bt * p.ncell;
    for (int i = 0; i <= k; i++) {
        ld[i * BS + tid] = 1e10f;   // knnquery_cuda_kernel.cu:88-91 fillers
        li[i * BS + tid] = start;
    }
    float worst = 1e10f;  // ld[k]
    int worst_i = 0x7fffffff;
    const int rmax = max(max(p.dims[0], p.dims[1]), p.dims[2]);
    for (int r = 0; r < rmax; r++) {
        for (int dz = -r; dz <= r; dz++) {
            const int z = cz + dz;
            if (z < 0 || z >= p.dims[2]) continue;
            for (int dy = -r; dy <= r; dy++) {
                const int y = cy + dy;
                if (y < 0 || y >= p.dims[1]) continue;
                const bool face = (dz == -r || dz == r || dy == -r || dy == r);
                // A face row of the shell is the whole x-run [cx-r, cx+r]: its cells are consecutive cell ids, so
                // their candidates are ONE contiguous range of records (two cell_start reads instead of 2r+1 dependent
                // pairs).  An interior row contributes only its two x-ends.  The order in which candidates are seen
                // does not matter: the list is ordered by (distance, index), exact ties are replayed.
                const int row = base + (z * p.dims[1] + y) * p.dims[0];
                const int nseg = (face || r == 0) ? 1 : 2;
                for (int seg = 0; seg < nseg; seg++) {
                    int x0, x1;
                    if (face || r == 0) { x0 = max(cx - r, 0); x1 = min(cx + r, p.dims[0] - 1); }
                    else { x0 = x1 = seg == 0 ? cx - r : cx + r; }
                    if (x0 < 0 || x1 >= p.dims[0] || x0 > x1) continue;
                    const int s = cell_start[row + x0], e = cell_start[row + x1 + 1];
                    for (int t = s; t < e; t++) {
                        const float4 c = rec[t];
                        const float ddx = qx - c.x, ddy = qy - c.y, ddz = qz - c.z;
                        const float d2 = __fmaf_rn(ddz, ddz, __fmaf_rn(ddx, ddx, __fmul_rn(ddy, ddy)));
                        const int ci = __float_as_int(c.w);
                        if (d2 < worst || (d2 == worst && ci < worst_i)) {
                            // insert into the ascending list (drop the old last entry)
                            int pos = k;
                            while (pos > 0) {
                                const float pd = ld[(pos - 1) * BS + tid];
                                const int pi = li[(pos - 1) * BS + tid];
                                if (pd < d2 || (pd == d2 && pi < ci)) break;
                                ld[pos * BS + tid] = pd;
                                li[pos * BS + tid] = pi;
                                pos--;
                            }
                            ld[pos * BS + tid] = d2;
                            li[pos * BS + tid] = ci;
                            worst = ld[k * BS + tid];
                            worst_i = worst < 1e10f ? li[k * BS + tid] : 0x7fffffff;
                        }
                    }
                }
            }
        }
        // nearest unvisited cell face (sides that have run out of grid do not bound anything)
        float dmin = INFINITY;
        bool more = false;
        if (cx - r > 0) { dmin = fminf(dmin, qx - (p.lo[0] + (cx - r) * p.c)); more = true; }
        if (cx + r < p.dims[0] - 1) { dmin = fminf(dmin, (p.lo[0] + (cx + r + 1) * p.c) - qx); more = true; }
        if (cy - r > 0) { dmin = fminf(dmin, qy - (p.lo[1] + (cy - r) * p.c)); more = true; }
        if (cy + r < p.dims[1] - 1) { dmin = fminf(dmin, (p.lo[1] + (cy + r + 1) * p.c) - qy); more = true; }
        if (cz - r > 0) { dmin = fminf(dmin, qz - (p.lo[2] + (cz - r) * p.c)); more = true; }
        if (cz + r < p.dims[2] - 1) { dmin = fminf(dmin, (p.lo[2] + (cz + r + 1) * p.c) - qz); more = true; }
        if (!more) break;
        dmin -= 1e-3f * p.c;  // fp32 cell assignment / face arithmetic slack
        if (dmin > 0.f && worst < dmin * dmin * 0.999f) break;
    }
    // exact ties among the k+1 best make the reference's order history-dependent: replay those queries
    bool tie = false;
    for (int i = 0; i < k; i++) {
        const float a = ld[i * BS + tid], bnext = ld[(i + 1) * BS + tid];
        if (a < 1e10f && a == bnext) tie = true;
    }
    if (tie) {
        replay[atomicAdd(replay_count, 1)] = pt;
        return;
    }
    for (int i = 0; i < k; i++) {
        idx[(size_t)pt * k + i] = li[i * BS + tid];
        dist2[(size_t)pt * k + i] = ld[i * BS + tid];
    }
}

// Several lanes per query.  The one-thread-per-query kernel above is a chain of dependent steps per candidate (record load,
// LDS list insertion; 64 unrelated insertion histories per wave), and 25 000 queries are 1.5 waves per CU.  Here LQ = 16 / 32 /
// 64 lanes share a query (k + 1 <= LQ): the group's best LQ candidates are ONE sorted register per lane (lane i of the group
// holds the i-th best as a 64-bit key, distance bits << 32 | index: the reference's (distance, index) order), candidates are
// taken LQ at a time with coalesced record loads (a batch is filled across the short row segments of a shell), a batch that
// holds nothing better than the (k+1)-th best is dropped after one vote, any other is sorted (bitonic network over lane
// exchanges) and merged (reverse + half-cleaners).  No LDS, no per-candidate chain; queries x LQ lanes fill the chip.
// Same shells, same stopping rule, same tie replay as above.
template <int LQ>
__device__ __forceinline__ unsigned long long group_exchange(unsigned long long v, int partner_xor) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, partner_xor, LQ), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), partner_xor, LQ);
    return ((unsigned long long)hi << 32) | lo;
}
template <int LQ>
__device__ __forceinline__ unsigned long long group_read(unsigned long long v, int src) {
    const unsigned lo = (unsigned)__shfl((int)(unsigned)v, src, LQ), hi = (unsigned)__shfl((int)(unsigned)(v >> 32), src, LQ);
    return ((unsigned long long)hi << 32) | lo;
}

template <int LQ>
__global__ __launch_bounds__(256) void knn_lanes_kernel(int m, int k, const float *__restrict__ new_xyz, const int *__restrict__ offset,
                                                        const int *__restrict__ new_offset, const KnnPlan *__restrict__ plan,
                                                        const int *__restrict__ cell_start, const float4 *__restrict__ rec,
                                                        int *__restrict__ idx, float *__restrict__ dist2, int *__restrict__ replay,
                                                        int *__restrict__ replay_count, const int *__restrict__ qperm) {
    const int tid = threadIdx.x, sub = tid & (LQ - 1);
    const int gq = (blockIdx.x * 256 + tid) / LQ;  // the group's query, in cell order (qperm)
    if (gq >= m) return;                           // (a whole group leaves; exchanges never cross groups)
    const int pt = qperm[gq];
    int bt = 0;
    while (!(pt < new_offset[bt])) bt++;
    const int start = bt == 0 ? 0 : offset[bt - 1];
    const KnnPlan p = *plan;
    const float qx = new_xyz[(size_t)pt * 3 + 0], qy = new_xyz[(size_t)pt * 3 + 1], qz = new_xyz[(size_t)pt * 3 + 2];
    const int cx = cell_coord(qx, p.lo[0], p.inv_c, p.dims[0]);
    const int cy = cell_coord(qy, p.lo[1], p.inv_c, p.dims[1]);
    const int cz = cell_coord(qz, p.lo[2], p.inv_c, p.dims[2]);
    const int base = bt * p.ncell;
    constexpr unsigned long long NONE = ~0ull;
    const unsigned long long filler = ((unsigned long long)__float_as_uint(1e10f) << 32) | (unsigned)start;  // knnquery_cuda_kernel.cu:88-91
    unsigned long long cur = filler;   // sorted ascending over the group's lanes
    unsigned long long worst = filler; // the (k+1)-th best: what a candidate has to beat
    unsigned long long nk = NONE;      // the batch being filled
    int fill = 0;                      // (group-uniform)
    const int gshift = (tid & 63) & ~(LQ - 1);
    const unsigned long long gmask = LQ == 64 ? ~0ull : (((1ull << LQ) - 1ull) << gshift);

    auto flush = [&]() {
        if (__ballot(nk < worst) & gmask) {
            // bitonic sort of the batch, ascending over the group
#pragma unroll
            for (int kk = 2; kk <= LQ; kk <<= 1)
#pragma unroll
                for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                    const unsigned long long o = group_exchange<LQ>(nk, jj);
                    const bool up = (sub & kk) == 0, lower = (sub & jj) == 0;
                    nk = (lower == up) ? (o < nk ? o : nk) : (o > nk ? o : nk);
                }
            // the LQ smallest of (cur ascending, batch ascending): minimum with the reversed batch is bitonic; clean it
            const unsigned long long rv = group_read<LQ>(nk, LQ - 1 - sub);
            unsigned long long mg = rv < cur ? rv : cur;
#pragma unroll
            for (int jj = LQ >> 1; jj > 0; jj >>= 1) {
                const unsigned long long o = group_exchange<LQ>(mg, jj);
                mg = ((sub & jj) == 0) ? (o < mg ? o : mg) : (o > mg ? o : mg);
            }
            cur = mg;
            worst = group_read<LQ>(cur, k);
        }
        nk = NONE;
        fill = 0;
    };
    auto add_segment = [&](int s, int e) {  // records [s, e) of the cell-ordered candidates
        while (s < e) {
            const int n = min(e - s, LQ - fill), rel = sub - fill;
            if (rel >= 0 && rel < n) {
                const float4 c = rec[s + rel];
                const float ddx = qx - c.x, ddy = qy - c.y, ddz = qz - c.z;
                const float d2 = __fmaf_rn(ddz, ddz, __fmaf_rn(ddx, ddx, __fmul_rn(ddy, ddy)));
                nk = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)__float_as_int(c.w);
            }
            fill += n;
            s += n;
            if (fill == LQ) flush();
        }
    };

    const int rmax = max(max(p.dims[0], p.dims[1]), p.dims[2]);
    for (int r = 0; r < rmax; r++) {
        for (int dz = -r; dz <= r; dz++) {
            const int z = cz + dz;
            if (z < 0 || z >= p.dims[2]) continue;
            for (int dy = -r; dy <= r; dy++) {
                const int y = cy + dy;
                if (y < 0 || y >= p.dims[1]) continue;
                const bool face = (dz == -r || dz == r || dy == -r || dy == r);
                const int row = base + (z * p.dims[1] + y) * p.dims[0];
                const int nseg = (face || r == 0) ? 1 : 2;
                for (int seg = 0; seg < nseg; seg++) {
                    int x0, x1;
                    if (face || r == 0) { x0 = max(cx - r, 0); x1 = min(cx + r, p.dims[0] - 1); }
                    else { x0 = x1 = seg == 0 ? cx - r : cx + r; }
                    if (x0 < 0 || x1 >= p.dims[0] || x0 > x1) continue;
                    add_segment(cell_start[row + x0], cell_start[row + x1 + 1]);
                }
            }
        }
        if (fill > 0) flush();
        const float wd = __uint_as_float((unsigned)(worst >> 32));
        // nearest unvisited cell face (sides that have run out of grid do not bound anything)
        float dmin = INFINITY;
        bool more = false;
        if (cx - r > 0) { dmin = fminf(dmin, qx - (p.lo[0] + (cx - r) * p.c)); more = true; }
        if (cx + r < p.dims[0] - 1) { dmin = fminf(dmin, (p.lo[0] + (cx + r + 1) * p.c) - qx); more = true; }
        if (cy - r > 0) { dmin = fminf(dmin, qy - (p.lo[1] + (cy - r) * p.c)); more = true; }
        if (cy + r < p.dims[1] - 1) { dmin = fminf(dmin, (p.lo[1] + (cy + r + 1) * p.c) - qy); more = true; }
        if (cz - r > 0) { dmin = fminf(dmin, qz - (p.lo[2] + (cz - r) * p.c)); more = true; }
        if (cz + r < p.dims[2] - 1) { dmin = fminf(dmin, (p.lo[2] + (cz + r + 1) * p.c) - qz); more = true; }
        if (!more) break;
        dmin -= 1e-3f * p.c;  // fp32 cell assignment / face arithmetic slack
        if (dmin > 0.f && wd < dmin * dmin * 0.999f) break;
    }
    // exact ties among the k+1 best make the reference's order history-dependent: replay those queries
    const unsigned long long nxt = group_read<LQ>(cur, min(sub + 1, LQ - 1));
    const float da = __uint_as_float((unsigned)(cur >> 32)), db = __uint_as_float((unsigned)(nxt >> 32));
    const bool tie = sub < k && da < 1e10f && da == db;
    if (__ballot(tie) & gmask) {
        if (sub == 0) replay[atomicAdd(replay_count, 1)] = pt;
        return;
    }
    if (sub < k) {
        idx[(size_t)pt * k + sub] = (int)(unsigned)cur;
        dist2[(size_t)pt * k + sub] = da;
    }
}

// literal heap procedure (knn.hip) for the listed queries only
__global__ __launch_bounds__(64) void knn_replay_kernel(int k, const int *__restrict__ replay, const int *__restrict__ replay_count,
                                                        const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                        const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                        int *__restrict__ idx, float *__restrict__ dist2) {
    constexpr int BS = 64;
    extern __shared__ float smem[];
    float *hd = smem;
    int *hi = reinterpret_cast<int *>(smem + (size_t)k * BS);
    const int tid = threadIdx.x;
    const int total = *replay_count;
    for (int w = blockIdx.x * BS + tid; w < total; w += gridDim.x * BS) {
        const int pt = replay[w];
        int bt = 0;
        while (!(pt < new_offset[bt])) bt++;
        const int start = bt == 0 ? 0 : offset[bt - 1], end = offset[bt];
        const float nx = new_xyz[(size_t)pt * 3 + 0], ny = new_xyz[(size_t)pt * 3 + 1], nz = new_xyz[(size_t)pt * 3 + 2];
        for (int i = 0; i < k; i++) { hd[i * BS + tid] = 1e10f; hi[i * BS + tid] = start; }
        auto reheap = [&](int kk) {
            int root = 0, child = 1;
            while (child < kk) {
                if (child + 1 < kk && hd[(child + 1) * BS + tid] > hd[child * BS + tid]) child++;
                const float dr = hd[root * BS + tid], dc = hd[child * BS + tid];
                if (dr > dc) return;
                hd[root * BS + tid] = dc; hd[child * BS + tid] = dr;
                const int ir = hi[root * BS + tid];
                hi[root * BS + tid] = hi[child * BS + tid]; hi[child * BS + tid] = ir;
                root = child;
                child = root * 2 + 1;
            }
        };
        float top = 1e10f;
        for (int i = start; i < end; i++) {
            const float dx = nx - xyz[(size_t)i * 3 + 0], dy = ny - xyz[(size_t)i * 3 + 1], dz = nz - xyz[(size_t)i * 3 + 2];
            const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
            if (d2 < top) {
                hd[tid] = d2; hi[tid] = i;
                reheap(k);
                top = hd[tid];
            }
        }
        for (int i = k - 1; i > 0; i--) {
            const float d0 = hd[tid]; hd[tid] = hd[i * BS + tid]; hd[i * BS + tid] = d0;
            const int i0 = hi[tid]; hi[tid] = hi[i * BS + tid]; hi[i * BS + tid] = i0;
            reheap(i);
        }
        for (int i = 0; i < k; i++) {
            idx[(size_t)pt * k + i] = hi[i * BS + tid];
            dist2[(size_t)pt * k + i] = hd[i * BS + tid];
        }
    }
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static int bits_for_cells(int b) {
    long long tot = (long long)KNN_CELL_CAP * b;
    int r = 1;
    while ((1ll << r) < tot) r++;
    return r;
}
static size_t knn_cub_bytes(int n, int b) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned *)nullptr, (unsigned *)nullptr, (const int *)nullptr,
                                             (int *)nullptr, n, 0, bits_for_cells(b), (hipStream_t) nullptr);
    return bytes;
}
static size_t knn_ws_bytes(int n, int m, int b) {
    return al(sizeof(KnnPlan)) + al((size_t)b * 24) + 4 * al((size_t)n * 4) + al((size_t)n * 16) +
           al(((size_t)KNN_CELL_CAP * b + 1) * 4) + al((size_t)m * 4) + al(256) + 4 * al((size_t)m * 4) +
           al(knn_cub_bytes(max(n, m), b));
}

// returns false when the grid path does not apply (caller runs the full scan)
bool knn_grid_launch(int m, int k, int n, int b, const float *xyz, const float *new_xyz, const int *offset,
                     const int *new_offset, int *idx, float *dist2) {
    Workspace &w = workspace();
    if (w.ptr == nullptr || n <= 0 || b <= 0 || (long long)KNN_CELL_CAP * b > (1ll << 30)) return false;
    if (w.bytes < knn_ws_bytes(n, m, b)) return false;
    if ((long long)m * n < (1ll << 22)) return false;  // small problems: the scan is as fast as building a grid
    hipStream_t st = state().stream;
    char *p = reinterpret_cast<char *>(w.ptr);
    KnnPlan *plan = (KnnPlan *)p; p += al(sizeof(KnnPlan));
    float *bbox = (float *)p; p += al((size_t)b * 24);
    unsigned *keys_in = (unsigned *)p; p += al((size_t)n * 4);
    unsigned *keys_out = (unsigned *)p; p += al((size_t)n * 4);
    int *vals_in = (int *)p; p += al((size_t)n * 4);
    int *vals_out = (int *)p; p += al((size_t)n * 4);
    float4 *rec = (float4 *)p; p += al((size_t)n * 16);
    int *cell_start = (int *)p; p += al(((size_t)KNN_CELL_CAP * b + 1) * 4);
    int *replay = (int *)p; p += al((size_t)m * 4);
    int *replay_count = (int *)p; p += al(256);
    unsigned *qkeys_in = (unsigned *)p; p += al((size_t)m * 4);
    unsigned *qkeys_out = (unsigned *)p; p += al((size_t)m * 4);
    int *qvals_in = (int *)p; p += al((size_t)m * 4);
    int *qperm = (int *)p; p += al((size_t)m * 4);
    void *cub_tmp = p;
    size_t cub_bytes = w.bytes - (size_t)(p - reinterpret_cast<char *>(w.ptr));

    (void)hipMemsetAsync(replay_count, 0, sizeof(int), st);
    launch_bbox(b, xyz, offset, bbox, st);
    hipLaunchKernelGGL(knn_plan_kernel, dim3(1), dim3(1), 0, st, b, n, bbox, plan);
    hipLaunchKernelGGL(knn_cell_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, n, b, xyz, offset, plan, keys_in, vals_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, (const unsigned *)keys_in, keys_out, (const int *)vals_in, vals_out,
                                                      n, 0, bits_for_cells(b), st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return true; }
    hipLaunchKernelGGL(knn_finish_kernel, dim3(div_up(n, 256)), dim3(256), 0, st, n, b, keys_out, vals_out, xyz, plan, cell_start, rec);
    // the queries in cell order (same grid, same key as the candidates)
    hipLaunchKernelGGL(knn_cell_kernel, dim3(div_up(m, 256)), dim3(256), 0, st, m, b, new_xyz, new_offset, plan, qkeys_in, qvals_in);
    e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, (const unsigned *)qkeys_in, qkeys_out, (const int *)qvals_in, qperm, m, 0,
                                           bits_for_cells(b), st);
    if (e != hipSuccess) { set_error(hipGetErrorString(e)); return true; }
    static const bool one_lane = getenv("P2_KNN_ONE_LANE") != nullptr;
    if (k + 1 <= 64 && !one_lane) {  // several lanes per query
#define P2_KNN_LANES(LQ_)                                                                                                              \
    hipLaunchKernelGGL(knn_lanes_kernel<LQ_>, dim3((unsigned)div_up64((int64_t)m * LQ_, 256)), dim3(256), 0, st, m, k, new_xyz, offset, new_offset, \
                       plan, cell_start, rec, idx, dist2, replay, replay_count, qperm)
        if (k + 1 <= 16) P2_KNN_LANES(16);
        else if (k + 1 <= 32) P2_KNN_LANES(32);
        else P2_KNN_LANES(64);
#undef P2_KNN_LANES
    } else {
        constexpr int BS = 64;
        const size_t lds = (size_t)(k + 1) * BS * 8;
        hipLaunchKernelGGL(knn_grid_kernel<BS>, dim3(div_up(m, BS)), dim3(BS), lds, st, m, k, new_xyz, offset, new_offset, plan, cell_start, rec,
                           idx, dist2, replay, replay_count, qperm);
    }
    hipLaunchKernelGGL(knn_replay_kernel, dim3(min(div_up(m, 64), 1024)), dim3(64), (size_t)k * 64 * 8, st, k, replay, replay_count, xyz, new_xyz,
                       offset, new_offset, idx, dist2);
    return true;
}

}  // namespace p2

using namespace p2;

extern "C" {

size_t pointops2_knn_workspace_bytes(int n, int m, int b) {
    if (n <= 0 || m <= 0 || b <= 0) return 0;
    return knn_ws_bytes(n, m, b);
}

}  // extern "C"
