// I1 (fast path, second generation): exact furthest point sampling in ROUNDS, gfx950.
//
// FPS is m dependent arg-max steps, and fps_bucket.hip pays ~1 us of one CU's latency for each of them.  But the
// steps are almost independent: a step only changes the running min-distance of the points NEAR its sample, and the
// points with the largest min-distances are spread over the whole cloud.  A round therefore decides MANY steps at once:
//
//   candidates  S = every point whose key (min-dist bits << 32 | tie rank) is >= a threshold tau; all other points
//               have keys < tau, now and - keys only decrease - for the rest of the round;
//   resolve     in key order (the order the reference would select them) a candidate j is the reference's next sample
//               iff no ALREADY ACCEPTED candidate i (key_i > key_j) lies closer to it than its min-distance
//               (d(i, j) < d_j: its key would have dropped first).  With the pairwise predicate
//               hit[j][i] = key_i > key_j && d(i, j) < d_j the accepted set is the unique fixed point of
//               acc_j = !exists i: hit[j][i] && acc_i  (a chain of dependencies along the key order; interactions are
//               rare, the iteration settles in a few sweeps).  A dropped candidate's new key is bounded by its
//               distance to the accepted candidates that hit it; nothing below the largest such bound (or below a
//               candidate that did not fit the list) is accepted in this round - it waits for the next one;
//   update      the accepted samples are applied to the cloud together: min-dist = min over the new samples, bucket by
//               bucket (points Morton-sorted in buckets of 64 with boxes; a bucket is skipped when no accepted sample
//               can reach it), with a two-level box test (16-bucket super-buckets first).
// The index sequence is the reference's, bit for bit (same fma chain, same tie ranks, min() is order-independent);
// ~170 rounds replace 25 000 dependent steps.  A round that accepts nothing (no candidate reached the threshold, or the
// candidate list overflowed above the best candidate) falls back to ONE literal step from the true maximum, so progress is
// unconditional.
//
// G workgroups of 16 waves per batch element (G = 1 ... 16 by the size of the cloud).  Workgroup g OWNS a contiguous range
// of buckets and keeps their points IN REGISTERS for the whole kernel (lane <-> point of a bucket, up to LZ_NSLOT buckets per
// wave): applying samples and gathering candidates - the part of a round that scales with the cloud - touches no memory.
// The resolve step is REPLICATED: the workgroups publish their candidates in global memory, meet at ONE grid barrier per
// round, and each of them then sorts and resolves the same list to the same accepted set (deterministic; only workgroup 0
// writes the indices).  The threshold of the next round is taken from a bound every workgroup knows (the previous threshold /
// the largest bound of what was not accepted) instead of the true maximum, which would need a second barrier; the true
// maximum arrives with the candidates and serves the literal-step fallback.
// State (pts.w = running min-dist, written back at the end; tie ranks; Morton order) and the resume / verified-prefix
// conventions are fps_bucket.hip's.
#include "fps_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace p2 {

constexpr int LZ_NW = 16;
constexpr int LZ_NT = LZ_NW * 64;
constexpr int LZ_WORDS = LZ_CAP / 32;
constexpr int LZ_TARGET = 352;           // candidates the threshold controller aims at (at most)
constexpr int LZ_GRID = 4096;            // hash slots of the candidate grid (1024: every thread walks ~5 foreign entries per round)
constexpr int LZ_HITS = 8;               // listed hitters per candidate
constexpr int LZ_PAIRS_MAX = 128;        // up to this many candidates the hitters are found by testing all pairs

// exchange area of one batch element (LZ_XCHG bytes, fps_common.h): barrier counter | headers [2][GMAX] | candidates [2][GMAX][CAP]
struct LzHdr {
    unsigned long long top;    // largest key among the workgroup's points (after its updates)
    unsigned long long tover;  // largest key of a candidate that did not fit the workgroup's list
    int count;                 // candidates found (may exceed LZ_CAP)
    int pad[3];
};
static_assert(sizeof(LzHdr) == 32, "header layout");
static_assert(LZ_XCHG >= 64 + 2 * LZ_GMAX * 32 + 2 * LZ_GMAX * LZ_CAP * 20, "exchange area too small");
static_assert(LZ_NT / LZ_GMAX >= 64, "a wave reads entries of one workgroup's list");

template <typename T>
__device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All G workgroups of a batch element arrive; every workgroup reaches it the same number of times (the control flow around
// it is replicated), so the grid always drains.  EVERY byte that crosses workgroups is written by a device-scope (sc1) store
// and read by a device-scope (sc1) load, every storing wave drains its stores (vmcnt(0)) before the workgroup's barrier, and
// the counter itself is a relaxed device-scope atomic: no release / acquire FENCE anywhere.  A fence would write back and
// invalidate the whole L2 of the workgroup's XCD - sixteen workgroups on eight XCDs, every ~35 us - and the attention kernels
// that run beside the sampler live on L2 hits: with release / acquire atomics here they took 1.6x (forward) to 2x (backward)
// as long (tools/interference.py).
// Safety net: a workgroup that has waited `patience` ticks of the 100 MHz clock (2 s: the whole kernel takes milliseconds)
// poisons the counter, which releases every waiter of the element, and all of them leave (returns false): a grid that cannot
// make progress for a reason outside the algorithm (its workgroups not resident together: CUs held by other work) must still
// drain.  The element's indices are then incomplete; the poisoning workgroup sets ASYNC_FPS_BARRIER_TIMEOUT in the library's
// status word, which pointops2_last_error() turns into an error at the next library call after the kernel ran.
constexpr unsigned LZ_POISON = 0x40000000u;
extern unsigned long long g_fps_patience;  // misc.hip (pointops2_diag_set_fps_patience): LZ_PATIENCE in ticks, 2 s by default
__device__ __forceinline__ bool group_barrier(unsigned *bar, unsigned target, int G, int *s_flag, unsigned long long patience, unsigned *status) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's device-scope stores have been performed
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned v = target;
        if (G > 1) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            while ((v = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > patience) {
                    __hip_atomic_fetch_add(bar, LZ_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // the call's indices are incomplete from here on: say so where the host will see it (common.h, async_status_word)
                    if (status != nullptr) __hip_atomic_store(status, ASYNC_FPS_BARRIER_TIMEOUT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (a plain store: no PCIe atomics needed)
                    v = LZ_POISON;
                    break;
                }
            }
        }
        *s_flag = v >= LZ_POISON ? 1 : 0;
    }
    __syncthreads();
    return *s_flag == 0;
}

constexpr size_t lz_lds_bytes() {
    return 16 * LZ_CAP + 8 * LZ_CAP + 8 * LZ_NW + 4 * LZ_CAP * 9 + 4 * LZ_GRID + 4 * LZ_NSLOT * LZ_WORDS + 4 * 2 * LZ_NSLOT + 4 * LZ_NSLOT * 6 + 4 * 2 * LZ_WORDS +
           2 * LZ_CAP * LZ_HITS + 2 * LZ_CAP * 2 + 4 * LZ_CAP + 16 * 24;  // (+ alignment slack)
}

__device__ __forceinline__ unsigned ord_bits(float v) {  // order-preserving float -> unsigned
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord_float(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
// lower bound of the squared distance from a point to a box: same fma chain as sqd on clamped differences, every
// rounding monotone, so lb <= d(point, p) in fp32 for every p inside the box (fps_bucket.hip)
__device__ __forceinline__ float box_lb(float x, float y, float z, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
    const float dx = fmaxf(fmaxf(mnx - x, x - mxx), 0.f);
    const float dy = fmaxf(fmaxf(mny - y, y - mxy), 0.f);
    const float dz = fmaxf(fmaxf(mnz - z, z - mxz), 0.f);
    return sqd(dx, dy, dz);
}
__device__ __forceinline__ unsigned cell_hash(int x, int y, int z) {
    return ((unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u) & (LZ_GRID - 1);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int l) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}

// STAMP: diagnostic build only (P2_FPS_STAMPS=1): cycle sums of the phases of wave 0 of workgroup 0 and round statistics -> dbg
template <bool STAMP, int NSLOT>
__global__ __launch_bounds__(LZ_NT) void fps_lazy_kernel(int Bref, int log2B, int bid0, const float *__restrict__ xyz, const int *__restrict__ offset,
                                                         const int *__restrict__ new_offset, float4 *__restrict__ pts,
                                                         const unsigned *__restrict__ rank, const int *__restrict__ prev_idx,
                                                         const int *__restrict__ prev_offset, const int *__restrict__ verified,
                                                         int *__restrict__ idx, unsigned char *__restrict__ xchg_all,
                                                         unsigned long long patience, unsigned *__restrict__ status,
                                                         unsigned long long *__restrict__ dbg = nullptr) {
    unsigned long long c_ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    auto stamp = [&](int ph) {
        if (STAMP) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            c_ph[ph] += t - t_last;
            t_last = t;
        }
    };
    if (STAMP) t_last = __builtin_amdgcn_s_memtime();
    constexpr int NW = LZ_NW, NT = LZ_NT, CAP = LZ_CAP, WORDS = LZ_WORDS;
    static_assert(NSLOT <= LZ_NSLOT, "LDS is sized for LZ_NSLOT super-buckets");
    // LDS: carved from one dynamic block (more than the 64 KiB a kernel may declare statically)
    extern __shared__ unsigned char lz_lds[];
    unsigned char *lp = lz_lds;
    auto carve = [&](size_t bytes) { unsigned char *r = lp; lp += (bytes + 15) & ~(size_t)15; return r; };
    float4 *sp4 = reinterpret_cast<float4 *>(carve(sizeof(float4) * CAP));                 // candidate (x, y, z, min-dist), by sorted position
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(carve(8 * CAP));     // candidate keys, sorted descending (the reference's selection order)
    unsigned long long *wkey = reinterpret_cast<unsigned long long *>(carve(8 * NW));
    float *cx = reinterpret_cast<float *>(carve(4 * CAP)), *cy = reinterpret_cast<float *>(carve(4 * CAP));  // candidates, in list order
    float *cz = reinterpret_cast<float *>(carve(4 * CAP)), *cd = reinterpret_cast<float *>(carve(4 * CAP));
    unsigned *clo = reinterpret_cast<unsigned *>(carve(4 * CAP));
    unsigned *ccell = reinterpret_cast<unsigned *>(carve(4 * CAP));                        // candidate's grid cell (10 bits per axis), by sorted position
    int *ghead = reinterpret_cast<int *>(carve(4 * LZ_GRID));                              // hash grid over the candidates: chains of sorted positions
    float *ax = reinterpret_cast<float *>(carve(4 * CAP)), *ay = reinterpret_cast<float *>(carve(4 * CAP));  // accepted samples, in selection order
    float *az = reinterpret_cast<float *>(carve(4 * CAP));
    unsigned (*sbhit)[WORDS] = reinterpret_cast<unsigned (*)[WORDS]>(carve(4 * LZ_NSLOT * WORDS));  // accepted samples that may reach a super-bucket
    unsigned (*sbmax)[LZ_NSLOT] = reinterpret_cast<unsigned (*)[LZ_NSLOT]>(carve(4 * 2 * LZ_NSLOT));      // largest min-dist (bits) inside a super-bucket, double-buffered
    unsigned (*sbbox)[6] = reinterpret_cast<unsigned (*)[6]>(carve(4 * LZ_NSLOT * 6));              // super-bucket boxes (ord_bits)
    unsigned (*accw)[WORDS] = reinterpret_cast<unsigned (*)[WORDS]>(carve(4 * 2 * WORDS));
    unsigned short (*hl)[LZ_HITS] = reinterpret_cast<unsigned short (*)[LZ_HITS]>(carve(2 * CAP * LZ_HITS));  // hitters of a candidate (sorted positions, all with larger keys)
    unsigned short *sidx = reinterpret_cast<unsigned short *>(carve(2 * CAP));             // sorted position -> list slot
    short *gnext = reinterpret_cast<short *>(carve(2 * CAP));
    int *hcnt = reinterpret_cast<int *>(carve(4 * CAP));                                   // number of hitters; > LZ_HITS: too many to list
    __shared__ unsigned long long s_tover, s_tdrop;
    __shared__ int s_cnt, s_nacc, s_changed[2], s_abort;

    // every exit below depends on the batch element alone: the G workgroups of an element leave together
    const int G = gridDim.x, g = blockIdx.x, bid = bid0 + blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    if (end_n <= start_n) {
        if (g == 0)
            for (int j = start_m + tid; j < end_m; j += NT) idx[j] = start_n;
        return;
    }
    const int n = end_n - start_n, m = end_m - start_m;
    const int nb = (n + 63) / 64;
    const int nbw = (nb + G - 1) / G;                         // buckets per workgroup (<= NW * NSLOT: the launcher's choice of G)
    const int b0 = g * nbw, nbl = max(0, min(nbw, nb - b0));  // this workgroup's buckets: [b0, b0 + nbl)
    const int nsb = (nbl + NW - 1) / NW;                      // super-buckets in use (<= NSLOT): the buckets of one register slot
    unsigned char *xchg = xchg_all + (size_t)bid * LZ_XCHG;
    unsigned *bar = reinterpret_cast<unsigned *>(xchg);
    LzHdr *hdr = reinterpret_cast<LzHdr *>(xchg + 64);                                         // [2][LZ_GMAX]
    float *xc4 = reinterpret_cast<float *>(xchg + 64 + 2 * LZ_GMAX * sizeof(LzHdr));         // [2][LZ_GMAX][CAP][4]
    unsigned *xlo = reinterpret_cast<unsigned *>(xchg + 64 + 2 * LZ_GMAX * sizeof(LzHdr) + (size_t)2 * LZ_GMAX * CAP * sizeof(float4));

    // samples inherited from the previous call on this state, or verified to be the identity prefix (fps_bucket.hip);
    // workgroup 0 writes them, everybody knows the last one
    int done = 0, first = start_n;
    if (prev_idx) {
        const int ps = bid == 0 ? 0 : prev_offset[bid - 1], pe = prev_offset[bid];
        done = min(pe - ps, m);
        if (g == 0)
            for (int t = tid; t < done; t += NT) idx[start_m + t] = prev_idx[ps + t];
        if (done > 0) first = prev_idx[ps + done - 1];
    }
    if (verified) {
        const int v = min(verified[bid], m);
        if (v > done) {
            if (g == 0)
                for (int t = tid; t < v; t += NT) idx[start_m + t] = start_n + t;
            done = v;
            first = start_n + v - 1;
        }
    }
    if (done >= m) return;

    // ---- owned buckets: register slot sl of this wave <-> local bucket sl * NW + wave; lane <-> point.  The bucket's box and
    //      largest key live in lane sl. ----
    float X[NSLOT], Y[NSLOT], Z[NSLOT], W[NSLOT];
    unsigned R[NSLOT];
    float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY, mxz = -INFINITY;  // (an absent bucket is never reached)
    unsigned long long key = 0ull;
    for (int t = tid; t < NSLOT; t += NT) {
        sbbox[t][0] = sbbox[t][1] = sbbox[t][2] = 0xffffffffu;
        sbbox[t][3] = sbbox[t][4] = sbbox[t][5] = 0u;
        sbmax[0][t] = sbmax[1][t] = 0u;
    }
    for (int t = tid; t < NSLOT * WORDS; t += NT) sbhit[t / WORDS][t % WORDS] = 0u;
    __syncthreads();
#pragma unroll
    for (int sl = 0; sl < NSLOT; sl++) {
        const int li = sl * NW + wave;
        X[sl] = Y[sl] = Z[sl] = W[sl] = 0.f;
        R[sl] = 0u;
        if (li < nbl) {  // wave-uniform
            const int pos = min(start_n + (b0 + li) * 64 + lane, end_n - 1);  // lanes past the end copy the last point
            const float4 p = pts[pos];
            X[sl] = p.x; Y[sl] = p.y; Z[sl] = p.z; W[sl] = p.w;
            R[sl] = rank[pos];
            float a0 = p.x, a1 = p.y, a2 = p.z, c0 = p.x, c1 = p.y, c2 = p.z;
            for (int st = 1; st < 64; st <<= 1) {
                a0 = fminf(a0, __shfl_xor(a0, st, 64)); a1 = fminf(a1, __shfl_xor(a1, st, 64)); a2 = fminf(a2, __shfl_xor(a2, st, 64));
                c0 = fmaxf(c0, __shfl_xor(c0, st, 64)); c1 = fmaxf(c1, __shfl_xor(c1, st, 64)); c2 = fmaxf(c2, __shfl_xor(c2, st, 64));
            }
            const KeyMax km = wave_key_max(((unsigned long long)__float_as_uint(p.w) << 32) | R[sl]);
            if (lane == sl) {
                mnx = a0; mny = a1; mnz = a2; mxx = c0; mxy = c1; mxz = c2;
                key = km.key;
            }
        }
    }
    const bool own = lane < NSLOT && lane * NW + wave < nbl;  // this lane holds a bucket's box and key
    if (own) {
        atomicMin(&sbbox[lane][0], ord_bits(mnx)); atomicMin(&sbbox[lane][1], ord_bits(mny)); atomicMin(&sbbox[lane][2], ord_bits(mnz));
        atomicMax(&sbbox[lane][3], ord_bits(mxx)); atomicMax(&sbbox[lane][4], ord_bits(mxy)); atomicMax(&sbbox[lane][5], ord_bits(mxz));
        atomicMax(&sbmax[0][lane], (unsigned)(key >> 32));
    }
    if (tid == 0 && done == 0 && g == 0) idx[start_m] = start_n;
    // the last selected sample has not been applied to the min-dist field yet (fps_bucket.hip's convention; applying a
    // sample twice is harmless): it is the first round's accepted set
    if (tid == 0) {
        ax[0] = xyz[(size_t)first * 3 + 0];
        ay[0] = xyz[(size_t)first * 3 + 1];
        az[0] = xyz[(size_t)first * 3 + 2];
    }
    done = max(done, 1);
    int A = 1;                                   // accepted samples waiting to be applied
    int target = 64, lastK = 0;                  // candidates the threshold controller aims at: grows while most candidates are accepted
    float frac = fminf(0.5f, 2.0f * 64 / (float)n);  // threshold = (bound of the top min-dist) * (1 - frac)
    float bnd = INFINITY;                        // upper bound of the largest min-dist every workgroup knows (inf: nothing gathered in the first round)
    int buf = 0;                                 // sbmax[buf] = current super-bucket maxima
    unsigned round = 0;
    __syncthreads();

    stamp(0);  // 0: set-up
    for (;;) {
        // ================= update: apply the A accepted samples =================
        const int nwA = (A + 31) >> 5;
        if (STAMP) { c_ph[12] += 1; c_ph[13] += A; }
        // (1) sample x super-bucket box tests -> sbhit (zero on entry); one (super-bucket, 8 samples) unit per thread trip
        {
            const int nu8 = (A + 7) >> 3;
            for (int u = tid; u < nsb * nu8; u += NT) {
                const int sb = u / nu8, w8 = u - sb * nu8;
                const float bx0 = ord_float(sbbox[sb][0]), by0 = ord_float(sbbox[sb][1]), bz0 = ord_float(sbbox[sb][2]);
                const float bx1 = ord_float(sbbox[sb][3]), by1 = ord_float(sbbox[sb][4]), bz1 = ord_float(sbbox[sb][5]);
                const float dmax = __uint_as_float(sbmax[buf][sb]);
                unsigned bits = 0u;
                const int a0 = w8 * 8, a1 = min(A, a0 + 8);
                for (int a = a0; a < a1; a++)
                    if (box_lb(ax[a], ay[a], az[a], bx0, by0, bz0, bx1, by1, bz1) < dmax) bits |= 1u << (a & 31);
                if (bits) atomicOr(&sbhit[sb][a0 >> 5], bits);
            }
        }
        if (tid < NSLOT) sbmax[buf ^ 1][tid] = 0u;
        __syncthreads();
        stamp(1);  // 1: sample x super-bucket tests
        // (2) lane sl: which of those samples reach its bucket (up to four ids kept; more: all of the super-bucket's)
        unsigned long long lst = 0ull;
        int cntl = 0;
        if (own) {
            const float dmax = __uint_as_float((unsigned)(key >> 32));
            for (int w = 0; w < nwA; w++) {
                unsigned bits = sbhit[lane][w];
                while (bits) {
                    const int a = w * 32 + __ffs(bits) - 1;
                    bits &= bits - 1;
                    if (box_lb(ax[a], ay[a], az[a], mnx, mny, mnz, mxx, mxy, mxz) < dmax) {
                        if (cntl < 4) lst |= (unsigned long long)a << (16 * cntl);
                        cntl++;
                    }
                }
            }
        }
        stamp(2);  // 2: own-bucket tests
        // (3) the wave updates its touched buckets in registers
        {
            const unsigned touched = (unsigned)__ballot(cntl > 0);  // (bits = slots, wave-uniform)
#pragma unroll
            for (int sl = 0; sl < NSLOT; sl++) {
                if (!((touched >> sl) & 1u)) continue;
                const int c = __builtin_amdgcn_readlane(cntl, sl);
                const unsigned long long ids = readlane64(lst, sl);
                float d2 = W[sl];
                if (c <= 4) {
                    for (int v = 0; v < c; v++) {
                        const int a = (int)((ids >> (16 * v)) & 0xffffu);
                        d2 = fminf(d2, sqd(X[sl] - ax[a], Y[sl] - ay[a], Z[sl] - az[a]));
                    }
                } else {
                    for (int w = 0; w < nwA; w++) {
                        unsigned bits = (unsigned)__builtin_amdgcn_readfirstlane((int)sbhit[sl][w]);
                        while (bits) {
                            const int a = w * 32 + __ffs(bits) - 1;
                            bits &= bits - 1;
                            d2 = fminf(d2, sqd(X[sl] - ax[a], Y[sl] - ay[a], Z[sl] - az[a]));
                        }
                    }
                }
                W[sl] = d2;
                const KeyMax km = wave_key_max(((unsigned long long)__float_as_uint(d2) << 32) | R[sl]);
                if (lane == sl) key = km.key;
            }
        }
        stamp(3);  // 3: bucket updates
        // ================= select =================
        {
            const KeyMax wm = wave_key_max(own ? key : 0ull);
            if (lane == 0) wkey[wave] = wm.key;
            if (own) atomicMax(&sbmax[buf ^ 1][lane], (unsigned)(key >> 32));
        }
        buf ^= 1;
        if (tid == 0) { s_cnt = 0; s_tover = 0ull; s_tdrop = 0ull; s_nacc = 0; s_changed[0] = 0; s_changed[1] = 0; }
        for (int t = tid; t < LZ_GRID; t += NT) ghead[t] = -1;
        for (int t = tid; t < CAP; t += NT) hcnt[t] = 0;
        __syncthreads();
        for (int t = tid; t < NSLOT * WORDS; t += NT) sbhit[t / WORDS][t % WORDS] = 0u;  // (last read above; written again after more barriers)
        stamp(4);  // 4: maxima + barrier (waiting for the slowest wave's updates)
        if (done >= m) break;  // (uniform, and the same in every workgroup) everything selected, and applied
        unsigned long long ltop = 0ull;  // this workgroup's largest key
        {
            const unsigned long long v = lane < NW ? wkey[lane] : 0ull;
            ltop = wave_key_max(v).key;
        }
        // ---- gather this workgroup's candidates: every point with min-dist >= tau -> its list in the exchange area ----
        const int par = round & 1;
        const float tau = fmaxf(bnd * (1.0f - frac), 0.f);
        const unsigned taub = __float_as_uint(tau);
        {
            float *my4 = xc4 + ((size_t)par * LZ_GMAX + g) * CAP * 4;
            unsigned *mylo = xlo + ((size_t)par * LZ_GMAX + g) * CAP;
            const unsigned have = (unsigned)__ballot(own && (unsigned)(key >> 32) >= taub);
#pragma unroll
            for (int sl = 0; sl < NSLOT; sl++) {
                if (!((have >> sl) & 1u)) continue;
                const bool cand = start_n + (b0 + sl * NW + wave) * 64 + lane < end_n && __float_as_uint(W[sl]) >= taub;
                const unsigned long long cm = __ballot(cand);
                if (cm) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_cnt, __popcll(cm));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (cand) {
                        const int at = base + __popcll(cm & ((1ull << lane) - 1ull));
                        if (at < CAP) {
                            st_agent(my4 + at * 4 + 0, X[sl]); st_agent(my4 + at * 4 + 1, Y[sl]);
                            st_agent(my4 + at * 4 + 2, Z[sl]); st_agent(my4 + at * 4 + 3, W[sl]);
                            st_agent(mylo + at, R[sl]);
                        } else {
                            atomicMax(&s_tover, ((unsigned long long)__float_as_uint(W[sl]) << 32) | R[sl]);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            LzHdr *h = hdr + par * LZ_GMAX + g;
            st_agent(&h->top, ltop);
            st_agent(&h->tover, (unsigned long long)s_tover);
            st_agent(&h->count, (int)s_cnt);
        }
        stamp(5);  // 5: gather
        round++;
        if (!group_barrier(bar, round * (unsigned)G, G, &s_abort, patience, status)) return;  // (uniform in the workgroup; see LZ_PATIENCE)
        stamp(10);  // 10: waiting for the other workgroups
        // ---- everybody reads everybody's list (in workgroup order, cut at CAP: what is cut is bounded by its workgroup's top).
        //      One trip: lane q of every wave fetches header q while the wave fetches - speculatively - entries of "its" workgroup ----
        int K = 0, found = 0;
        unsigned long long gtop = 0ull, tover = 0ull;
        {
            const int E = NT / G;                       // list entries a pass covers per workgroup (>= 64; G divides NT)
            const int gg = tid / E, e0 = tid - gg * E;  // (gg is wave-uniform)
            unsigned long long h_top = 0ull, h_tov = 0ull;
            int h_cnt = 0;
            if (lane < G) {
                const LzHdr *h = hdr + par * LZ_GMAX + lane;
                h_top = ld_agent(&h->top); h_tov = ld_agent(&h->tover); h_cnt = ld_agent(&h->count);
            }
            const size_t seg = ((size_t)par * LZ_GMAX + gg) * CAP;
            // two entries per thread in flight with the headers: 2 E entries of every list arrive in the first trip
            float sx[2] = {0.f, 0.f}, sy[2] = {0.f, 0.f}, sz[2] = {0.f, 0.f}, sd[2] = {0.f, 0.f};
            unsigned slo[2] = {0u, 0u};
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = e0 + u * E;
                if (e < CAP) {
                    sx[u] = ld_agent(xc4 + (seg + e) * 4 + 0); sy[u] = ld_agent(xc4 + (seg + e) * 4 + 1); sz[u] = ld_agent(xc4 + (seg + e) * 4 + 2);
                    sd[u] = ld_agent(xc4 + (seg + e) * 4 + 3);
                    slo[u] = ld_agent(xlo + seg + e);
                }
            }
            int mybase = 0, mytake = 0, maxtake = 0;
            for (int q = 0; q < G; q++) {  // (uniform)
                const int c = __builtin_amdgcn_readlane(h_cnt, q);
                const unsigned long long tp = readlane64(h_top, q), tv = readlane64(h_tov, q);
                found += c;
                gtop = tp > gtop ? tp : gtop;
                tover = tv > tover ? tv : tover;
                const int keep = min(c, CAP), take = min(keep, CAP - K);
                if (take < keep) tover = tp > tover ? tp : tover;
                if (q == gg) { mybase = K; mytake = take; }
                maxtake = max(maxtake, take);
                K += take;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int e = e0 + u * E;
                if (e < mytake) { cx[mybase + e] = sx[u]; cy[mybase + e] = sy[u]; cz[mybase + e] = sz[u]; cd[mybase + e] = sd[u]; clo[mybase + e] = slo[u]; }
            }
            for (int e = e0 + 2 * E; e - e0 < maxtake; e += E) {  // (uniform trip count) the rare longer list
                if (e < mytake) {
                    cx[mybase + e] = ld_agent(xc4 + (seg + e) * 4 + 0); cy[mybase + e] = ld_agent(xc4 + (seg + e) * 4 + 1);
                    cz[mybase + e] = ld_agent(xc4 + (seg + e) * 4 + 2); cd[mybase + e] = ld_agent(xc4 + (seg + e) * 4 + 3);
                    clo[mybase + e] = ld_agent(xlo + seg + e);
                }
            }
        }
        const float dtop = __uint_as_float((unsigned)(gtop >> 32));
        {
            // threshold controller (every thread of every workgroup computes the same): aim at `target` candidates
            const float ratio = fminf(fmaxf((float)target / (float)max(found, 1), 0.5f), 2.0f);
            if (found > CAP) frac = fmaxf(frac * 0.25f * (float)CAP / (float)found, 1e-7f);
            else frac = fminf(fmaxf(frac * ratio, 1e-7f), 0.5f);
        }
        __syncthreads();
        stamp(11);  // 11: reading the lists
        lastK = K;
        // ---- order the candidates by key, descending: sorted position = the order the reference would select them in.  Keys are
        //      unique, so the position of a candidate is the number of larger keys: counted by all threads (NT / KP of them share a
        //      candidate, each over a contiguous slice of the list; a wave reads the same key at a time: LDS broadcasts) ----
        {
            const int KP = max((K + 63) & ~63, 64), S = NT / KP;
            const int part = tid / KP, j = tid - part * KP;
            unsigned long long *ukey = reinterpret_cast<unsigned long long *>(sp4);  // (unsorted keys: sp4 is written after the sort)
            if (tid < K) {
                ukey[tid] = ((unsigned long long)__float_as_uint(cd[tid]) << 32) | clo[tid];
                hcnt[tid] = 0;  // (doubles as the position counter; zeroed again below)
            }
            __syncthreads();
            if (part < S && j < K) {
                const unsigned long long mine = ukey[j];
                const int per = (K + S - 1) / S, i0 = part * per, i1 = min(K, i0 + per);
                int cnt = 0, i = i0;
                for (; i + 8 <= i1; i += 8) {  // eight keys in flight: the loop is LDS latency otherwise
                    unsigned long long kk[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) kk[u] = ukey[i + u];
#pragma unroll
                    for (int u = 0; u < 8; u++) cnt += kk[u] > mine ? 1 : 0;
                }
                for (; i < i1; i++) cnt += ukey[i] > mine ? 1 : 0;
                if (cnt) atomicAdd(&hcnt[j], cnt);
            }
            __syncthreads();
            if (tid < K) {
                const int pos = hcnt[tid];
                skey[pos] = ukey[tid];
                sidx[pos] = (unsigned short)tid;
            }
            __syncthreads();
            if (tid < K) hcnt[tid] = 0;
        }
        stamp(6);  // 6: sort
        // ---- hitters.  hit(r, r2) = r2 < r (larger key) && d(r2, r) < d_r.  Few candidates: all pairs, spread over the whole
        //      workgroup.  Many: a hash grid - a hit needs |p_r - p_r2| < sqrt(d_top) per axis, so with cells of that edge (plus
        //      slack for the rounding of the cell index; cell indices wrap at 1024, which only merges cells) only the 27 cells
        //      around a candidate can hold a hitter; the 27 cells of a candidate are split over the threads that share it ----
        float px = 0.f, py = 0.f, pz = 0.f, pd = 0.f;
        if (tid < K) {
            const int c = sidx[tid];
            px = cx[c]; py = cy[c]; pz = cz[c]; pd = cd[c];
            sp4[tid] = make_float4(px, py, pz, pd);
        }
        if (K <= LZ_PAIRS_MAX) {
            __syncthreads();
            const int j = tid & (LZ_PAIRS_MAX - 1), part = tid / LZ_PAIRS_MAX;  // NT / PAIRS_MAX = 8 threads share candidate j
            if (j < K) {
                const float4 me = sp4[j];
                for (int r2 = part; r2 < j; r2 += NT / LZ_PAIRS_MAX) {
                    const float4 q = sp4[r2];
                    if (sqd(me.x - q.x, me.y - q.y, me.z - q.z) < me.w) {
                        const int at = atomicAdd(&hcnt[j], 1);
                        if (at < LZ_HITS) hl[j][at] = (unsigned short)r2;
                    }
                }
            }
        } else {
            const float cell = sqrtf(dtop) * 1.002f + 1e-30f, inv_cell = 1.0f / cell;
            if (tid < K) {
                const int gx = (int)floorf(px * inv_cell) & 1023;
                const int gy = (int)floorf(py * inv_cell) & 1023;
                const int gz = (int)floorf(pz * inv_cell) & 1023;
                ccell[tid] = (unsigned)gx | ((unsigned)gy << 10) | ((unsigned)gz << 20);
                gnext[tid] = (short)atomicExch(&ghead[cell_hash(gx, gy, gz)], tid);
            }
            __syncthreads();
            stamp(14);  // 14: grid build (of rounds that use the grid)
            const int KP = (K + 63) & ~63, S = NT / KP;  // S = 2 ... 5 threads share a candidate (K > 128)
            const int part = tid / KP, j = tid - part * KP;
            if (part < S && j < K) {
                const float4 me = sp4[j];
                const unsigned cc = ccell[j];
                const int jx = cc & 1023, jy = (cc >> 10) & 1023, jz = cc >> 20;
                // the heads of all cells of this thread first (independent reads, one latency), then the chains
                int hd[14];  // (S >= 2: at most 14 of the 27 cells)
#pragma unroll
                for (int u = 0; u < 14; u++) {
                    const int c9 = part + u * S;
                    const int qx = (jx + c9 % 3 - 1) & 1023, qy = (jy + (c9 / 3) % 3 - 1) & 1023, qz = (jz + c9 / 9 - 1) & 1023;
                    hd[u] = c9 < 27 ? ghead[cell_hash(qx, qy, qz)] : -1;
                }
#pragma unroll
                for (int u = 0; u < 14; u++) {
                    const int c9 = part + u * S;
                    const int qx = (jx + c9 % 3 - 1) & 1023, qy = (jy + (c9 / 3) % 3 - 1) & 1023, qz = (jz + c9 / 9 - 1) & 1023;
                    const unsigned want = (unsigned)qx | ((unsigned)qy << 10) | ((unsigned)qz << 20);
                    for (int r2 = hd[u]; r2 >= 0;) {
                        const unsigned oc = ccell[r2];
                        const float4 q = sp4[r2];
                        const int nx = gnext[r2];
                        if (r2 < j && oc == want && sqd(me.x - q.x, me.y - q.y, me.z - q.z) < me.w) {
                            const int at = atomicAdd(&hcnt[j], 1);
                            if (at < LZ_HITS) hl[j][at] = (unsigned short)r2;
                        }
                        r2 = nx;
                    }
                }
            }
            stamp(15);  // 15: grid walk
        }
        if (tid < WORDS) {
            const int lo = tid * 32;
            accw[0][tid] = K >= lo + 32 ? 0xffffffffu : (K > lo ? ((1u << (K - lo)) - 1u) : 0u);
        }
        __syncthreads();
        const int nh = tid < K ? hcnt[tid] : 0;
        stamp(7);  // 7: hitters
        // ---- resolve: fixed point of acc_r = !exists r2 in hitters(r): acc_r2  (a candidate with more hitters than the list holds
        //      is simply not decided in this round) ----
        int cur = 0;
        for (int it = 0; it < CAP; it++) {
            bool mine = false, old = false;
            if (tid < K) {
                mine = nh <= LZ_HITS;
                for (int t = 0; t < min(nh, LZ_HITS); t++) {
                    const int r2 = hl[tid][t];
                    mine = mine && !((accw[cur][r2 >> 5] >> (r2 & 31)) & 1u);
                }
                old = (accw[cur][tid >> 5] >> (tid & 31)) & 1u;
            }
            const unsigned long long bm = __ballot(mine);
            if (lane == 0 && wave * 2 < WORDS) {
                accw[cur ^ 1][wave * 2] = (unsigned)bm;
                accw[cur ^ 1][wave * 2 + 1] = (unsigned)(bm >> 32);
            }
            if (mine != old) s_changed[it & 1] = 1;
            if (tid == 0) s_changed[(it + 1) & 1] = 0;
            __syncthreads();
            cur ^= 1;
            if (!s_changed[it & 1]) break;  // uniform: read after the barrier, reset two iterations later
        }
        stamp(8);  // 8: fixed point
        // ---- what was not accepted bounds what may be: a dropped candidate's new key is at most its key at the distance to
        //      the accepted candidates that hit it; an undecided one keeps its key ----
        bool acc = false;
        unsigned long long kj = 0ull;
        if (tid < K) {
            acc = (accw[cur][tid >> 5] >> (tid & 31)) & 1u;
            kj = skey[tid];
            if (!acc) {
                float nd = pd;
                bool any = false;
                for (int t = 0; t < min(nh, LZ_HITS); t++) {
                    const int r2 = hl[tid][t];
                    if ((accw[cur][r2 >> 5] >> (r2 & 31)) & 1u) {
                        const float4 q = sp4[r2];
                        nd = fminf(nd, sqd(px - q.x, py - q.y, pz - q.z));
                        any = true;
                    }
                }
                if (nh > LZ_HITS || !any) nd = pd;
                atomicMax(&s_tdrop, ((unsigned long long)__float_as_uint(nd) << 32) | (unsigned)kj);
            }
        }
        __syncthreads();
        const unsigned long long tcut = tover > s_tdrop ? tover : s_tdrop;
        {
            const int remaining = m - done;
            const bool fin0 = acc && kj > tcut;  // keys descend with the position: the accepted set is a prefix of the acc set
            const unsigned long long fm0 = __ballot(fin0);
            if (lane == 0 && wave * 2 < WORDS) {
                accw[cur ^ 1][wave * 2] = (unsigned)fm0;
                accw[cur ^ 1][wave * 2 + 1] = (unsigned)(fm0 >> 32);
            }
            __syncthreads();
            int r = 0;  // accepted candidates in front of this one = its place in the selection order
            if (fin0) {
                for (int w = 0; w < (tid >> 5); w++) r += __popc(accw[cur ^ 1][w]);
                r += __popc(accw[cur ^ 1][tid >> 5] & ((1u << (tid & 31)) - 1u));
            }
            const bool fin = fin0 && r < remaining;
            if (fin) {
                ax[r] = px; ay[r] = py; az[r] = pz;
                if (g == 0) idx[start_m + done + r] = start_n + rel_of(kj, Bref, log2B);
            }
            const unsigned long long fm = __ballot(fin);
            if (lane == 0 && fm) atomicAdd(&s_nacc, __popcll(fm));
        }
        __syncthreads();
        A = s_nacc;
        // few candidates while they get in each other's way (the early rounds), many once most of them are accepted
        if (A * 2 > lastK) target = min(target + target / 2, LZ_TARGET);
        else if (A * 4 < lastK) target = max(target / 2, 32);
        // what is left is bounded: points that were no candidates lie below tau, candidates that were not accepted at or below
        // the cut, and nothing exceeds the maximum the lists came with
        bnd = fminf(dtop, fmaxf(tau, __uint_as_float((unsigned)(tcut >> 32))));
        if (A == 0) {
            // nothing could be decided (no candidate reached the threshold - the first round, or a bound far above the true
            // maximum - or the list overflowed above the best candidate): one literal step from the true maximum
            if (tid == 0) {
                const int w = start_n + rel_of(gtop, Bref, log2B);
                ax[0] = xyz[(size_t)w * 3 + 0]; ay[0] = xyz[(size_t)w * 3 + 1]; az[0] = xyz[(size_t)w * 3 + 2];
                if (g == 0) idx[start_m + done] = w;
            }
            A = 1;
            bnd = dtop;
            __syncthreads();
        }
        if (STAMP && dbg && tid == 0 && g == 0 && round <= 512) dbg[16 + round - 1] = ((unsigned long long)found << 40) | ((unsigned long long)lastK << 20) | (unsigned)A;
        done += A;
        stamp(9);  // 9: bounds, ranks, output
    }
    // the state a resumed call (or fps_bucket.hip) continues from
#pragma unroll
    for (int sl = 0; sl < NSLOT; sl++) {
        const int pos = start_n + (b0 + sl * NW + wave) * 64 + lane;
        if (sl * NW + wave < nbl && pos < end_n) reinterpret_cast<float *>(pts + pos)[3] = W[sl];
    }
    if (STAMP && dbg && tid == 0 && g == 0)
        for (int i = 0; i < 16; i++) dbg[bid * 16 + i] = c_ph[i];
}

// Workgroups per batch element: enough that a workgroup's buckets fit its registers (NW * NSLOT), and about a hundred
// buckets each beyond that (more workgroups = less to do per round for each; the barrier costs the same).  0: the cloud is
// too large for this kernel.
int fps_lazy_groups(int n_max) {
    static const int env = getenv("P2_FPS_GROUPS") ? atoi(getenv("P2_FPS_GROUPS")) : 0;
    const int nb = (n_max + 63) / 64, cap = LZ_NW * LZ_NSLOT;
    const int need = (nb + cap - 1) / cap;
    if (need > LZ_GMAX) return 0;
    int G = env > 0 ? env : (nb + 99) / 100;
    G = std::max(std::max(need, 1), std::min(G, LZ_GMAX));
    while (G < LZ_GMAX && LZ_NT % G != 0) G++;  // the list readers want G to divide the workgroup: 1, 2, 4, 8, 16
    return G;
}

void fps_lazy_launch(int b, int n_max, int Bref, int log2B, const float *xyz, const int *offset, const int *new_offset, float4 *pts, const unsigned *rank,
                     const int *prev_idx, const int *prev_offset, const int *verified, int *idx, void *xchg, hipStream_t st) {
    const int G = fps_lazy_groups(n_max);
    const int slots = div_up(div_up(div_up(n_max, 64), G), LZ_NW);  // register slots a wave needs
    (void)hipMemsetAsync(xchg, 0, (size_t)b * LZ_XCHG, st);          // barrier counters (and headers)
    if (getenv("P2_FPS_STAMPS")) {  // diagnostic only: synchronous, prints the phase cycles of wave 0 of workgroup 0 to stderr
        unsigned long long *dbg = nullptr, host[16 + 512];  // (per-round trace of batch element 0 only)
        (void)hipMalloc(&dbg, sizeof(host) * b);
        (void)hipMemset(dbg, 0, sizeof(host) * b);
        if (slots <= 8) {
            allow_big_lds(fps_lazy_kernel<true, 8>, lz_lds_bytes());
            hipLaunchKernelGGL((fps_lazy_kernel<true, 8>), dim3(G, b), dim3(LZ_NT), lz_lds_bytes(), st, Bref, log2B, 0, xyz, offset, new_offset, pts, rank, prev_idx,
                               prev_offset, verified, idx, (unsigned char *)xchg, g_fps_patience, async_status_word(), dbg);
        } else {
            allow_big_lds(fps_lazy_kernel<true, LZ_NSLOT>, lz_lds_bytes());
            hipLaunchKernelGGL((fps_lazy_kernel<true, LZ_NSLOT>), dim3(G, b), dim3(LZ_NT), lz_lds_bytes(), st, Bref, log2B, 0, xyz, offset, new_offset, pts, rank, prev_idx,
                               prev_offset, verified, idx, (unsigned char *)xchg, g_fps_patience, async_status_word(), dbg);
        }
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        fprintf(stderr, "[fps lazy] G %d slots %d rounds %llu samples %llu | cycles: setup %llu sbtests %llu owntests %llu updates %llu maxima+wait %llu gather %llu gridwait %llu "
                        "lists %llu sort %llu hitters %llu (grid build %llu walk %llu) fixedpoint %llu output %llu\n", G, slots, host[12], host[13], host[0], host[1], host[2], host[3],
                host[4], host[5], host[10], host[11], host[6], host[7] + host[14] + host[15], host[14], host[15], host[8], host[9]);
        if (getenv("P2_FPS_TRACE_ROUNDS")) {
            fprintf(stderr, "[fps lazy] found/listed/accepted per round:");
            for (int i = 0; i < 512 && host[16 + i]; i++)
                fprintf(stderr, " %llu/%llu/%llu", host[16 + i] >> 40, (host[16 + i] >> 20) & 0xfffff, host[16 + i] & 0xfffff);
            fprintf(stderr, "\n");
        }
        return;
    }
    // the workgroups of an element wait for each other: all of a launch must be resident together (one workgroup per CU) -> at
    // most half the device's CUs per launch
    const int chunk = std::max(1, std::min(128, num_cus() / 2) / G);
    for (int c0 = 0; c0 < b; c0 += chunk) {
        const dim3 grid(G, std::min(chunk, b - c0));
        if (slots <= 8) {
            allow_big_lds(fps_lazy_kernel<false, 8>, lz_lds_bytes());
            hipLaunchKernelGGL((fps_lazy_kernel<false, 8>), grid, dim3(LZ_NT), lz_lds_bytes(), st, Bref, log2B, c0, xyz, offset, new_offset, pts, rank, prev_idx,
                               prev_offset, verified, idx, (unsigned char *)xchg, g_fps_patience, async_status_word(), (unsigned long long *)nullptr);
        } else {
            allow_big_lds(fps_lazy_kernel<false, LZ_NSLOT>, lz_lds_bytes());
            hipLaunchKernelGGL((fps_lazy_kernel<false, LZ_NSLOT>), grid, dim3(LZ_NT), lz_lds_bytes(), st, Bref, log2B, c0, xyz, offset, new_offset, pts, rank, prev_idx,
                               prev_offset, verified, idx, (unsigned char *)xchg, g_fps_patience, async_status_word(), (unsigned long long *)nullptr);
        }
        held_cus_note(st, (int)(grid.x * grid.y));
    }
}

}  // namespace p2
