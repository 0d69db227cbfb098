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
//               bucket as in fps_bucket.hip (points Morton-sorted in buckets of 64 with boxes; a bucket is skipped when
//               no accepted sample can reach it), with a two-level box test (16-bucket super-buckets first).
// The index sequence is the reference's, bit for bit (same fma chain, same tie ranks, min() is order-independent);
// ~100 rounds replace 25 000 dependent steps.  A round that accepts nothing (possible only when the candidate list
// overflowed) falls back to ONE literal step from the bucket maxima, so progress is unconditional.
//
// G workgroups of 16 waves per batch element (G = 1 ... 8 by the size of the cloud).  Workgroup g OWNS a contiguous range of
// buckets: it applies the accepted samples to them and gathers their candidates - the part of a round that scales with
// the cloud.  The resolve step is REPLICATED: the workgroups publish their candidates in global memory, meet at ONE grid
// barrier per round, and each of them then sorts and resolves the same list to the same accepted set (deterministic; only
// workgroup 0 writes the indices).  The threshold of the next round is taken from a bound every workgroup knows (the
// previous threshold / the largest bound of what was not accepted) instead of the true maximum, which would need a second
// barrier; the true maximum arrives with the candidates and serves the literal-step fallback.
// State (pts.w = running min-dist, tie ranks, Morton order) and the resume / verified-prefix conventions are fps_bucket.hip's.
#include "fps_common.h"
#include <cstdio>
#include <cstdlib>

namespace p2 {

constexpr int LZ_NW = 16;
constexpr int LZ_NT = LZ_NW * 64;
constexpr int LZ_NBL = 2;                // owned buckets per lane: up to 2 * 64 * 16 = 2048 buckets of 64 points
constexpr int LZ_WORDS = LZ_CAP / 32;
constexpr int LZ_TARGET = 352;           // candidates the threshold controller aims at (at most)
constexpr int LZ_MAXSB = LZ_NBL * 64;    // super-buckets: the 16 buckets (one per wave) with the same (slot, lane)
constexpr int LZ_GRID = 1024;            // hash cells of the candidate grid
constexpr int LZ_HITS = 8;               // listed hitters per candidate

// exchange area of one batch element (LZ_XCHG bytes, fps_common.h): barrier counter | headers [2][GMAX] | candidates [2][GMAX][CAP]
struct LzHdr {
    unsigned long long top;    // largest key among the workgroup's points (after its updates)
    unsigned long long tover;  // largest key of a candidate that did not fit the workgroup's list
    int count;                 // candidates found (may exceed LZ_CAP)
    int pad[3];
};
static_assert(sizeof(LzHdr) == 32, "header layout");
static_assert(LZ_XCHG >= 64 + 2 * LZ_GMAX * 32 + 2 * LZ_GMAX * LZ_CAP * 20, "exchange area too small");

template <typename T>
__device__ __forceinline__ T ld_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void st_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all G workgroups of a batch element arrive; every wave of the grid reaches it the same number of times (the control flow
// around it is replicated), so the grid always drains
// Safety net: a workgroup that has waited LZ_PATIENCE ticks of the 100 MHz clock (2 s: the whole kernel takes milliseconds)
// poisons the counter, which releases every waiter of the element, and all of them leave (returns false): a grid that cannot
// make progress for a reason outside the algorithm must still drain.
constexpr unsigned LZ_POISON = 0x40000000u;
constexpr unsigned long long LZ_PATIENCE = 200000000ull;
__device__ __forceinline__ bool group_barrier(unsigned *bar, unsigned target, int G, int *s_flag) {
    __threadfence();  // release: this thread's global stores are visible device-wide before the arrival
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned v = target;
        if (G > 1) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();
            while ((v = __hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (wall_clock64() - t0 > LZ_PATIENCE) {
                    __hip_atomic_fetch_add(bar, LZ_POISON, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    v = LZ_POISON;
                    break;
                }
            }
        }
        *s_flag = v >= LZ_POISON ? 1 : 0;
    }
    __syncthreads();
    __threadfence();  // acquire side for every thread
    return *s_flag == 0;
}

constexpr size_t lz_lds_bytes() {
    return 16 * LZ_CAP + 8 * LZ_CAP + 8 * LZ_NW + 4 * LZ_CAP * 9 + 4 * LZ_GRID + 4 * LZ_MAXSB * LZ_WORDS + 4 * 2 * LZ_MAXSB + 4 * LZ_MAXSB * 6 + 4 * 2 * LZ_WORDS +
           2 * LZ_CAP * LZ_HITS + 2 * LZ_CAP * 2 + LZ_CAP + 16 * 24;  // (+ alignment slack)
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

// STAMP: diagnostic build only (P2_FPS_STAMPS=1): cycle sums of the phases of wave 0 and round statistics -> dbg
template <bool STAMP>
__global__ __launch_bounds__(LZ_NT) void fps_lazy_kernel(int Bref, int log2B, const float *__restrict__ xyz, const int *__restrict__ offset,
                                                         const int *__restrict__ new_offset, float4 *__restrict__ pts,
                                                         const unsigned *__restrict__ rank, const int *__restrict__ prev_idx,
                                                         const int *__restrict__ prev_offset, const int *__restrict__ verified,
                                                         int *__restrict__ idx, unsigned char *__restrict__ xchg_all,
                                                         unsigned long long *__restrict__ dbg = nullptr) {
    unsigned long long c_ph[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;
    auto stamp = [&](int ph) {
        if (STAMP) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            c_ph[ph] += t - t_last;
            t_last = t;
        }
    };
    if (STAMP) t_last = __builtin_amdgcn_s_memtime();
    constexpr int NW = LZ_NW, NT = LZ_NT, NBL = LZ_NBL, CAP = LZ_CAP, WORDS = LZ_WORDS;
    // LDS: carved from one dynamic block (more than the 64 KiB a kernel may declare statically)
    extern __shared__ unsigned char lz_lds[];
    unsigned char *lp = lz_lds;
    auto carve = [&](size_t bytes) { unsigned char *r = lp; lp += (bytes + 15) & ~(size_t)15; return r; };
    float4 *sp4 = reinterpret_cast<float4 *>(carve(sizeof(float4) * CAP));                 // candidate (x, y, z, min-dist), by sorted position
    unsigned long long *skey = reinterpret_cast<unsigned long long *>(carve(8 * CAP));     // candidate keys, sorted descending (the reference's selection order)
    unsigned long long *wkey = reinterpret_cast<unsigned long long *>(carve(8 * NW));
    float *cx = reinterpret_cast<float *>(carve(4 * CAP)), *cy = reinterpret_cast<float *>(carve(4 * CAP));  // candidates, in gather order
    float *cz = reinterpret_cast<float *>(carve(4 * CAP)), *cd = reinterpret_cast<float *>(carve(4 * CAP));
    unsigned *clo = reinterpret_cast<unsigned *>(carve(4 * CAP));
    unsigned *ccell = reinterpret_cast<unsigned *>(carve(4 * CAP));                        // candidate's grid cell (10 bits per axis), by sorted position
    int *ghead = reinterpret_cast<int *>(carve(4 * LZ_GRID));                              // hash grid over the candidates: chains of sorted positions
    float *ax = reinterpret_cast<float *>(carve(4 * CAP)), *ay = reinterpret_cast<float *>(carve(4 * CAP));  // accepted samples, in selection order
    float *az = reinterpret_cast<float *>(carve(4 * CAP));
    unsigned (*sbhit)[WORDS] = reinterpret_cast<unsigned (*)[WORDS]>(carve(4 * LZ_MAXSB * WORDS));  // accepted samples that may reach a super-bucket
    unsigned (*sbmax)[LZ_MAXSB] = reinterpret_cast<unsigned (*)[LZ_MAXSB]>(carve(4 * 2 * LZ_MAXSB));  // largest min-dist (bits) inside a super-bucket, double-buffered
    unsigned (*sbbox)[6] = reinterpret_cast<unsigned (*)[6]>(carve(4 * LZ_MAXSB * 6));     // super-bucket boxes (ord_bits)
    unsigned (*accw)[WORDS] = reinterpret_cast<unsigned (*)[WORDS]>(carve(4 * 2 * WORDS));
    unsigned short (*hl)[LZ_HITS] = reinterpret_cast<unsigned short (*)[LZ_HITS]>(carve(2 * CAP * LZ_HITS));  // hitters of a candidate (sorted positions, all with larger keys)
    unsigned short *sidx = reinterpret_cast<unsigned short *>(carve(2 * CAP));             // sorted position -> candidate slot
    short *gnext = reinterpret_cast<short *>(carve(2 * CAP));
    unsigned char *hcnt = carve(CAP);                                                      // number of hitters; > LZ_HITS: too many to list
    __shared__ unsigned long long s_tover, s_tdrop;
    __shared__ int s_cnt, s_nacc, s_changed[2], s_abort;
    __shared__ unsigned s_org[3];                           // cloud origin (ord_bits of the bounding box minimum)

    // every exit below depends on the batch element alone: the G workgroups of an element leave together
    const int G = gridDim.x, g = blockIdx.x, bid = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    if (end_n <= start_n) {
        if (g == 0)
            for (int j = start_m + tid; j < end_m; j += NT) idx[j] = start_n;
        return;
    }
    const int n = end_n - start_n, m = end_m - start_m;
    const int nb = (n + 63) / 64;
    const int nbw = (nb + G - 1) / G;                      // buckets per workgroup
    const int b0 = g * nbw, nbl = max(0, min(nbw, nb - b0));  // this workgroup's buckets: [b0, b0 + nbl)
    const int nsb = (nbl + NW - 1) / NW;                   // super-buckets in use (<= LZ_MAXSB)
    unsigned char *xchg = xchg_all + (size_t)bid * LZ_XCHG;
    unsigned *bar = reinterpret_cast<unsigned *>(xchg);
    LzHdr *hdr = reinterpret_cast<LzHdr *>(xchg + 64);                                         // [2][LZ_GMAX]
    float4 *xc4 = reinterpret_cast<float4 *>(xchg + 64 + 2 * LZ_GMAX * sizeof(LzHdr));       // [2][LZ_GMAX][CAP]
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

    // ---- owned buckets: slot s of lane l  <->  bucket (s*64 + l)*NW + wave; box and largest key in registers ----
    float mnx[NBL], mny[NBL], mnz[NBL], mxx[NBL], mxy[NBL], mxz[NBL];
    unsigned long long key[NBL];
#pragma unroll
    for (int s = 0; s < NBL; s++) {
        mnx[s] = mny[s] = mnz[s] = INFINITY;  // an absent bucket is never reached and never holds a candidate
        mxx[s] = mxy[s] = mxz[s] = -INFINITY;
        key[s] = 0ull;
    }
    for (int t = tid; t < LZ_MAXSB; t += NT) {
        sbbox[t][0] = sbbox[t][1] = sbbox[t][2] = 0xffffffffu;
        sbbox[t][3] = sbbox[t][4] = sbbox[t][5] = 0u;
        sbmax[0][t] = sbmax[1][t] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < NBL; s++) {
        for (int l = 0; l < 64; l++) {
            const int li = (s * 64 + l) * NW + wave;
            if (li >= nbl) break;
            const int pos = min(start_n + (b0 + li) * 64 + lane, end_n - 1);
            const float4 p = pts[pos];
            float a0 = p.x, a1 = p.y, a2 = p.z, b0 = p.x, b1 = p.y, b2 = p.z;
            for (int st = 1; st < 64; st <<= 1) {
                a0 = fminf(a0, __shfl_xor(a0, st, 64)); a1 = fminf(a1, __shfl_xor(a1, st, 64)); a2 = fminf(a2, __shfl_xor(a2, st, 64));
                b0 = fmaxf(b0, __shfl_xor(b0, st, 64)); b1 = fmaxf(b1, __shfl_xor(b1, st, 64)); b2 = fmaxf(b2, __shfl_xor(b2, st, 64));
            }
            const KeyMax km = wave_key_max(((unsigned long long)__float_as_uint(p.w) << 32) | rank[pos]);
            if (lane == l) {
                mnx[s] = a0; mny[s] = a1; mnz[s] = a2; mxx[s] = b0; mxy[s] = b1; mxz[s] = b2;
                key[s] = km.key;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < NBL; s++)
        if (key[s] != 0ull || mnx[s] != INFINITY) {
            const int sb = s * 64 + lane;
            atomicMin(&sbbox[sb][0], ord_bits(mnx[s])); atomicMin(&sbbox[sb][1], ord_bits(mny[s])); atomicMin(&sbbox[sb][2], ord_bits(mnz[s]));
            atomicMax(&sbbox[sb][3], ord_bits(mxx[s])); atomicMax(&sbbox[sb][4], ord_bits(mxy[s])); atomicMax(&sbbox[sb][5], ord_bits(mxz[s]));
            atomicMax(&sbmax[0][sb], (unsigned)(key[s] >> 32));
        }
    if (tid == 0 && done == 0 && g == 0) idx[start_m] = start_n;
    __syncthreads();  // boxes complete
    // the last selected sample has not been applied to the min-dist field yet (fps_bucket.hip's convention; applying a
    // sample twice is harmless): it is the first round's accepted set
    {
        if (tid == 0) {
            ax[0] = xyz[(size_t)first * 3 + 0];
            ay[0] = xyz[(size_t)first * 3 + 1];
            az[0] = xyz[(size_t)first * 3 + 2];
        }
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
        // (1) sample x super-bucket box tests -> sbhit; one (super-bucket, 32 samples) unit per thread trip
        for (int u = tid; u < nsb * nwA; u += NT) {
            const int sb = u / nwA, w = u - sb * nwA;
            const float bx0 = ord_float(sbbox[sb][0]), by0 = ord_float(sbbox[sb][1]), bz0 = ord_float(sbbox[sb][2]);
            const float bx1 = ord_float(sbbox[sb][3]), by1 = ord_float(sbbox[sb][4]), bz1 = ord_float(sbbox[sb][5]);
            const float dmax = __uint_as_float(sbmax[buf][sb]);
            unsigned bits = 0u;
            const int a0 = w * 32, a1 = min(A, a0 + 32);
            for (int a = a0; a < a1; a++)
                if (box_lb(ax[a], ay[a], az[a], bx0, by0, bz0, bx1, by1, bz1) < dmax) bits |= 1u << (a - a0);
            sbhit[sb][w] = bits;
        }
        for (int t = tid; t < LZ_MAXSB; t += NT) sbmax[buf ^ 1][t] = 0u;
        __syncthreads();
        stamp(1);  // 1: sample x super-bucket tests
        // (2) every lane: which of those samples reach its own buckets (up to four ids kept; more: all of the super-bucket's)
        unsigned long long lst[NBL];
        int cntl[NBL];
#pragma unroll
        for (int s = 0; s < NBL; s++) {
            lst[s] = 0ull;
            cntl[s] = 0;
            const int sb = s * 64 + lane;
            if (sb < nsb && mnx[s] != INFINITY) {
                const float dmax = __uint_as_float((unsigned)(key[s] >> 32));
                for (int w = 0; w < nwA; w++) {
                    unsigned bits = sbhit[sb][w];
                    while (bits) {
                        const int a = w * 32 + __ffs(bits) - 1;
                        bits &= bits - 1;
                        if (box_lb(ax[a], ay[a], az[a], mnx[s], mny[s], mnz[s], mxx[s], mxy[s], mxz[s]) < dmax) {
                            if (cntl[s] < 4) lst[s] |= (unsigned long long)a << (16 * cntl[s]);
                            cntl[s]++;
                        }
                    }
                }
            }
        }
        stamp(2);  // 2: own-bucket tests
        // (3) the wave updates its touched buckets: 64 lanes <-> 64 points, the loads of up to four buckets in flight together
#pragma unroll
        for (int s = 0; s < NBL; s++) {
            unsigned long long touched = __ballot(cntl[s] > 0);
            while (touched) {
                int ol[4], pos[4];
                float4 p[4];
                unsigned rk[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ol[u] = touched ? __ffsll(touched) - 1 : -1;  // wave-uniform
                    touched &= touched - 1;                       // (0 & anything stays 0)
                    pos[u] = min(start_n + (b0 + (s * 64 + max(ol[u], 0)) * NW + wave) * 64 + lane, end_n - 1);  // lanes past the end copy the last point
                    if (ol[u] >= 0) { p[u] = pts[pos[u]]; rk[u] = rank[pos[u]]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (ol[u] < 0) continue;  // wave-uniform
                    const int c = __builtin_amdgcn_readlane(cntl[s], ol[u]);
                    const unsigned l0 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)lst[s], ol[u]);
                    const unsigned l1 = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(lst[s] >> 32), ol[u]);
                    float d2 = p[u].w;
                    if (c <= 4) {
                        const unsigned long long ids = ((unsigned long long)l1 << 32) | l0;
                        for (int v = 0; v < c; v++) {
                            const int a = (int)((ids >> (16 * v)) & 0xffffu);
                            d2 = fminf(d2, sqd(p[u].x - ax[a], p[u].y - ay[a], p[u].z - az[a]));
                        }
                    } else {
                        const int sb = s * 64 + ol[u];
                        for (int w = 0; w < nwA; w++) {
                            unsigned bits = (unsigned)__builtin_amdgcn_readfirstlane((int)sbhit[sb][w]);
                            while (bits) {
                                const int a = w * 32 + __ffs(bits) - 1;
                                bits &= bits - 1;
                                d2 = fminf(d2, sqd(p[u].x - ax[a], p[u].y - ay[a], p[u].z - az[a]));
                            }
                        }
                    }
                    reinterpret_cast<float *>(pts + pos[u])[3] = d2;
                    const KeyMax km = wave_key_max(((unsigned long long)__float_as_uint(d2) << 32) | rk[u]);
                    if (lane == ol[u]) key[s] = km.key;
                }
            }
        }
        stamp(3);  // 3: bucket updates
        // ================= select =================
        {
            unsigned long long mk = key[0];
#pragma unroll
            for (int s = 1; s < NBL; s++) mk = key[s] > mk ? key[s] : mk;
            const KeyMax wm = wave_key_max(mk);
            if (lane == 0) wkey[wave] = wm.key;
#pragma unroll
            for (int s = 0; s < NBL; s++)
                if (mnx[s] != INFINITY) atomicMax(&sbmax[buf ^ 1][s * 64 + lane], (unsigned)(key[s] >> 32));
        }
        buf ^= 1;
        if (tid == 0) { s_cnt = 0; s_tover = 0ull; s_tdrop = 0ull; s_nacc = 0; s_changed[0] = 0; s_changed[1] = 0; }
        if (tid < 3) s_org[tid] = 0xffffffffu;
        for (int t = tid; t < LZ_GRID; t += NT) ghead[t] = -1;
        __syncthreads();
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
            float4 *my4 = xc4 + ((size_t)par * LZ_GMAX + g) * CAP;
            unsigned *mylo = xlo + ((size_t)par * LZ_GMAX + g) * CAP;
#pragma unroll
            for (int s = 0; s < NBL; s++) {
                unsigned long long have = __ballot(mnx[s] != INFINITY && (unsigned)(key[s] >> 32) >= taub);
                while (have) {
                    int ol[4], pos[4];
                    float4 p[4];
                    unsigned rk[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        ol[u] = have ? __ffsll(have) - 1 : -1;
                        have &= have - 1;
                        pos[u] = start_n + (b0 + (s * 64 + max(ol[u], 0)) * NW + wave) * 64 + lane;
                        if (ol[u] >= 0) { p[u] = pts[min(pos[u], end_n - 1)]; rk[u] = rank[min(pos[u], end_n - 1)]; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        if (ol[u] < 0) continue;  // wave-uniform
                        const bool cand = pos[u] < end_n && __float_as_uint(p[u].w) >= taub;
                        const unsigned long long cm = __ballot(cand);
                        if (cm) {
                            int base = 0;
                            if (lane == 0) base = atomicAdd(&s_cnt, __popcll(cm));
                            base = __builtin_amdgcn_readfirstlane(base);
                            if (cand) {
                                const int at = base + __popcll(cm & ((1ull << lane) - 1ull));
                                if (at < CAP) {
                                    my4[at] = p[u];
                                    mylo[at] = rk[u];
                                } else {
                                    atomicMax(&s_tover, ((unsigned long long)__float_as_uint(p[u].w) << 32) | rk[u]);
                                }
                            }
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
        if (!group_barrier(bar, round * (unsigned)G, G, &s_abort)) return;  // (uniform in the workgroup; see LZ_PATIENCE)
        stamp(10);  // 10: waiting for the other workgroups
        // ---- everybody reads everybody's list (in workgroup order, cut at CAP: what is cut is bounded by its workgroup's top) ----
        int K = 0, found = 0;
        unsigned long long gtop = 0ull, tover = 0ull;
        int gbase[LZ_GMAX + 1];
#pragma unroll
        for (int gg = 0; gg < LZ_GMAX; gg++) {
            gbase[gg] = K;
            if (gg < G) {
                const LzHdr *h = hdr + par * LZ_GMAX + gg;
                const unsigned long long tp = ld_agent(&h->top), tv = ld_agent(&h->tover);
                const int c = ld_agent(&h->count);
                found += c;
                gtop = tp > gtop ? tp : gtop;
                tover = tv > tover ? tv : tover;
                const int keep = min(c, CAP), take = min(keep, CAP - K);
                if (take < keep) tover = tp > tover ? tp : tover;
                K += take;
            }
        }
        gbase[LZ_GMAX] = K;
        for (int t = tid; t < K; t += NT) {
            int gg = 0;
#pragma unroll
            for (int q = 1; q < LZ_GMAX; q++) gg += (q < G && t >= gbase[q]) ? 1 : 0;
            int gb = 0;
#pragma unroll
            for (int q = 0; q < LZ_GMAX; q++) gb = q == gg ? gbase[q] : gb;
            const size_t at = ((size_t)par * LZ_GMAX + gg) * CAP + (t - gb);
            const float *src = reinterpret_cast<const float *>(xc4 + at);
            const float x = ld_agent(src + 0), y = ld_agent(src + 1), z = ld_agent(src + 2), d = ld_agent(src + 3);
            cx[t] = x; cy[t] = y; cz[t] = z; cd[t] = d;
            clo[t] = ld_agent(xlo + at);
            atomicMin(&s_org[0], ord_bits(x)); atomicMin(&s_org[1], ord_bits(y)); atomicMin(&s_org[2], ord_bits(z));
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
        // ---- sort the candidates by key, descending: sorted position = the order the reference would select them in ----
        int P2 = 2;
        while (P2 < K) P2 <<= 1;
        if (tid < P2) {
            skey[tid] = tid < K ? (((unsigned long long)__float_as_uint(cd[tid]) << 32) | clo[tid]) : 0ull;
            sidx[tid] = (unsigned short)tid;
        }
        __syncthreads();
        for (int kk = 2; kk <= P2; kk <<= 1)
            for (int jj = kk >> 1; jj > 0; jj >>= 1) {
                const int other = tid ^ jj;
                if (tid < P2 && other > tid) {
                    const unsigned long long a = skey[tid], b = skey[other];
                    if ((a < b) == ((tid & kk) == 0)) {  // descending overall
                        skey[tid] = b; skey[other] = a;
                        const unsigned short t0 = sidx[tid];
                        sidx[tid] = sidx[other]; sidx[other] = t0;
                    }
                }
                __syncthreads();
            }
        stamp(6);  // 6: sort
        // ---- hitters through a hash grid.  hit(r, r2) = r2 < r (larger key) && d(r2, r) < d_r needs |p_r - p_r2| < sqrt(d_top) per
        //      axis: with cells of that edge (plus slack for the rounding of the cell index) only the 27 cells around a
        //      candidate can hold a hitter ----
        const float cell = sqrtf(dtop) * 1.002f + 1e-30f, inv_cell = 1.0f / cell;
        const float ox = ord_float(s_org[0]), oy = ord_float(s_org[1]), oz = ord_float(s_org[2]);
        float px = 0.f, py = 0.f, pz = 0.f, pd = 0.f;
        int gx = 0, gy = 0, gz = 0;
        auto cell_hash = [](int x, int y, int z) -> unsigned {
            return ((unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u) & (LZ_GRID - 1);
        };
        if (tid < K) {
            const int c = sidx[tid];
            px = cx[c]; py = cy[c]; pz = cz[c]; pd = cd[c];
            gx = min(max((int)floorf((px - ox) * inv_cell), 0), 1022);  // (clamped cells only merge: still a superset)
            gy = min(max((int)floorf((py - oy) * inv_cell), 0), 1022);
            gz = min(max((int)floorf((pz - oz) * inv_cell), 0), 1022);
            ccell[tid] = (unsigned)gx | ((unsigned)gy << 10) | ((unsigned)gz << 20);
            sp4[tid] = make_float4(px, py, pz, pd);
            gnext[tid] = (short)atomicExch(&ghead[cell_hash(gx, gy, gz)], tid);
        }
        __syncthreads();
        int nh = 0;
        if (tid < K) {
            for (int dz = -1; dz <= 1; dz++)
                for (int dy = -1; dy <= 1; dy++)
                    for (int dx = -1; dx <= 1; dx++) {
                        const int qx = gx + dx, qy = gy + dy, qz = gz + dz;
                        if (qx < 0 || qy < 0 || qz < 0) continue;
                        const unsigned want = (unsigned)qx | ((unsigned)qy << 10) | ((unsigned)qz << 20);
                        for (int r2 = ghead[cell_hash(qx, qy, qz)]; r2 >= 0; r2 = gnext[r2]) {
                            if (r2 >= tid || ccell[r2] != want) continue;  // smaller key, or another cell of the same hash chain
                            const float4 q = sp4[r2];
                            if (sqd(px - q.x, py - q.y, pz - q.z) < pd) {
                                if (nh < LZ_HITS) hl[tid][nh] = (unsigned short)r2;
                                nh++;
                            }
                        }
                    }
            hcnt[tid] = (unsigned char)min(nh, 255);
        }
        if (tid < WORDS) {
            const int lo = tid * 32;
            accw[0][tid] = K >= lo + 32 ? 0xffffffffu : (K > lo ? ((1u << (K - lo)) - 1u) : 0u);
        }
        __syncthreads();
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
        done += A;
        stamp(9);  // 9: bounds, ranks, output
    }
    if (STAMP && dbg && tid == 0 && g == 0)
        for (int i = 0; i < 14; i++) dbg[bid * 14 + i] = c_ph[i];
}

// workgroups per batch element: a workgroup should own ~100+ buckets for its share of a round to outweigh the barrier;
// all b * G workgroups must be resident together (they wait for each other), so large batches get fewer
static int lz_groups(int b, int n_max) {
    static const int env = getenv("P2_FPS_GROUPS") ? atoi(getenv("P2_FPS_GROUPS")) : 0;
    const int nb = (n_max + 63) / 64;
    int G = env > 0 ? env : nb / 96;
    G = std::max(1, std::min(G, LZ_GMAX));
    while (G > 1 && b * G > 64) G--;
    return G;
}

void fps_lazy_launch(int b, int n_max, int Bref, int log2B, const float *xyz, const int *offset, const int *new_offset, float4 *pts, const unsigned *rank,
                     const int *prev_idx, const int *prev_offset, const int *verified, int *idx, void *xchg, hipStream_t st) {
    allow_big_lds(fps_lazy_kernel<false>, lz_lds_bytes());
    allow_big_lds(fps_lazy_kernel<true>, lz_lds_bytes());
    const int G = lz_groups(b, n_max);
    (void)hipMemsetAsync(xchg, 0, (size_t)b * LZ_XCHG, st);  // barrier counters (and headers)
    if (getenv("P2_FPS_STAMPS")) {  // diagnostic only: synchronous, prints the phase cycles of wave 0 of workgroup 0 to stderr
        unsigned long long *dbg = nullptr, host[14];
        (void)hipMalloc(&dbg, sizeof(host) * b);
        (void)hipMemset(dbg, 0, sizeof(host) * b);
        hipLaunchKernelGGL(fps_lazy_kernel<true>, dim3(G, b), dim3(LZ_NT), lz_lds_bytes(), st, Bref, log2B, xyz, offset, new_offset, pts, rank, prev_idx, prev_offset,
                           verified, idx, (unsigned char *)xchg, dbg);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
        (void)hipFree(dbg);
        fprintf(stderr, "[fps lazy] G %d rounds %llu samples %llu | cycles: setup %llu sbtests %llu owntests %llu updates %llu maxima+wait %llu gather %llu gridwait %llu "
                        "lists %llu sort %llu hitters %llu fixedpoint %llu output %llu\n", G, host[12], host[13], host[0], host[1], host[2], host[3], host[4], host[5],
                host[10], host[11], host[6], host[7], host[8], host[9]);
        return;
    }
    hipLaunchKernelGGL(fps_lazy_kernel<false>, dim3(G, b), dim3(LZ_NT), lz_lds_bytes(), st, Bref, log2B, xyz, offset, new_offset, pts, rank, prev_idx, prev_offset,
                       verified, idx, (unsigned char *)xchg, (unsigned long long *)nullptr);
}

}  // namespace p2
