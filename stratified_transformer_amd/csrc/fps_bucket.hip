// I1 (fast path): exact furthest point sampling with spatial buckets, gfx950.
//
// FPS is m dependent arg-max steps.  The reference (and sampling.hip) rescans the whole cloud in
// every step: 20*N bytes and ~N/1024 loop trips per step from one workgroup.  Almost all of that
// work cannot change anything: a point's running min-distance only drops if the new sample is
// closer to it than its current value.  This kernel keeps the SAME arithmetic and the SAME winner
// (bit-exact index sequence, including the reference's launch-geometry tie rule, see sampling.hip)
// but skips, per step, every bucket of points that provably cannot change:
//
//   set-up  points of each batch element are sorted along a Morton (z-order) curve and cut into
//           buckets of BSZ consecutive points (BSZ = 64 for n <= 131072); a bucket has an axis-
//           aligned box and a cached maximum  key = (min-dist bits << 32 | tie rank)  of its points.
//   step    (a) every thread tests buckets: lower bound lb of the squared distance from the new
//               sample to the box, computed with the SAME fma chain as the point distance — every
//               rounding in that chain is monotone, so lb <= d(sample, p) in fp32 for every p in
//               the box; if lb >= cached max min-dist, min(d, tmp) == tmp for the whole bucket: skip;
//           (b) touched buckets go to an LDS work list; one wave per bucket updates its points
//               (coalesced, L2-resident), re-reduces the bucket key and stores the arg-max's
//               coordinates next to it;
//           (c) the global arg-max is a reduction over the <= 2048 bucket keys in LDS; the winner's
//               coordinates come from LDS too, so a step has no dependent global load besides (b).
//   Late in the sampling a step touches a handful of buckets (~1 wave pass); the whole sampling is
//   ~m * 3 workgroup barriers instead of m full-cloud scans.
//
// One workgroup per batch element (as the reference): no cross-workgroup synchronisation at all.
#include "common.h"
#include <hipcub/hipcub.hpp>

namespace p2 {

struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
};
Workspace &workspace() {
    static thread_local Workspace w;
    return w;
}

constexpr int FPS_MAX_BUCKETS = 2048;

__device__ __forceinline__ float sqd(float dx, float dy, float dz) {
    return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
}
// wave-wide unsigned max with DPP row operations (6 VALU + 1 readlane; a __shfl_xor butterfly costs 6
// dependent ds_bpermute round trips, and the 64-bit key would double that)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_step(unsigned v) {
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false);
    return v > t ? v : t;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    v = dpp_step<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v = dpp_step<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v = dpp_step<0x141, 0xF>(v);  // row_half_mirror
    v = dpp_step<0x140, 0xF>(v);  // row_mirror: every lane of a 16-lane row holds the row max
    v = dpp_step<0x142, 0xA>(v);  // row_bcast15 -> rows 1,3
    v = dpp_step<0x143, 0xC>(v);  // row_bcast31 -> rows 2,3: lane 63 holds the wave max
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// max of (hi:lo) keys: max hi first, then max lo among the lanes that hold it
__device__ __forceinline__ unsigned long long wmax64(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = wave_max_u32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}
__device__ __forceinline__ unsigned long long key_of(float d2, int rel, int Bref, int log2B) {
    const unsigned tref = (unsigned)rel & (unsigned)(Bref - 1);
    const unsigned cidx = (unsigned)rel >> log2B;
    const unsigned brev = log2B ? (__brev(tref) >> (32 - log2B)) : 0u;
    return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(0x7fffffffu - ((brev << 21) | cidx));
}
__device__ __forceinline__ int rel_of(unsigned long long key, int Bref, int log2B) {
    const unsigned key2 = 0x7fffffffu - (unsigned)(key & 0xffffffffull);
    const unsigned brev = key2 >> 21, cidx = key2 & ((1u << 21) - 1);
    const unsigned tref = log2B ? (__brev(brev) >> (32 - log2B)) : 0u;
    return (int)(cidx * (unsigned)Bref + tref);
}

// ---- set-up ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fps_bbox_kernel(const float *__restrict__ xyz, const int *__restrict__ offset,
                                                       float *__restrict__ bbox) {
    __shared__ float red[6][4];
    const int bid = blockIdx.x, tid = threadIdx.x;
    const int s = bid == 0 ? 0 : offset[bid - 1], e = offset[bid];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = s + tid; i < e; i += 256)
        for (int a = 0; a < 3; a++) {
            const float v = xyz[(size_t)i * 3 + a];
            mn[a] = fminf(mn[a], v);
            mx[a] = fmaxf(mx[a], v);
        }
    for (int a = 0; a < 3; a++) {
        for (int st = 1; st < 64; st <<= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], st, 64));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], st, 64));
        }
        if ((tid & 63) == 0) { red[a][tid >> 6] = mn[a]; red[3 + a][tid >> 6] = mx[a]; }
    }
    __syncthreads();
    if (tid < 3) bbox[bid * 6 + tid] = fminf(fminf(red[tid][0], red[tid][1]), fminf(red[tid][2], red[tid][3]));
    else if (tid < 6) bbox[bid * 6 + tid] = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
}

__device__ __forceinline__ unsigned spread10(unsigned v) {  // 10 bits -> every third bit
    v &= 0x3ff;
    v = (v | (v << 16)) & 0x030000ff;
    v = (v | (v << 8)) & 0x0300f00f;
    v = (v | (v << 4)) & 0x030c30c3;
    v = (v | (v << 2)) & 0x09249249;
    return v;
}

__global__ void fps_morton_kernel(int N, int b, const float *__restrict__ xyz, const int *__restrict__ offset,
                                  const float *__restrict__ bbox, unsigned long long *__restrict__ keys, int *__restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int bid = 0;
    while (bid < b - 1 && i >= offset[bid]) bid++;
    unsigned code = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = bbox[bid * 6 + a], hi = bbox[bid * 6 + 3 + a];
        const float ext = fmaxf(hi - lo, 1e-20f);
        int c = (int)((xyz[(size_t)i * 3 + a] - lo) / ext * 1024.f);
        c = min(max(c, 0), 1023);
        code |= spread10((unsigned)c) << a;
    }
    keys[i] = ((unsigned long long)bid << 32) | code;
    vals[i] = i;
}

// sorted-order point records: (x, y, z, running min-dist) as one 16-byte load, plus the 31-bit tie rank
__global__ void fps_gather_kernel(int N, int b, int Bref, int log2B, const float *__restrict__ xyz, const int *__restrict__ offset,
                                  const int *__restrict__ order, float4 *__restrict__ pts, unsigned *__restrict__ rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int o = order[i];
    pts[i] = make_float4(xyz[(size_t)o * 3 + 0], xyz[(size_t)o * 3 + 1], xyz[(size_t)o * 3 + 2], 1e10f);  // pointops.py:26
    int bid = 0;
    while (bid < b - 1 && i >= offset[bid]) bid++;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    rank[i] = (unsigned)key_of(0.f, o - start_n, Bref, log2B);
}

// ---- sampling --------------------------------------------------------------------------------
// Two-level hierarchy: buckets of BSZ points, super-buckets of 32 consecutive buckets (<= 64 supers, one
// per lane).  Four waves (one per SIMD, so no wave shares a SIMD's issue slots); per step:
//   S1  every wave (redundantly, no barrier): lane s tests super box s          -> 64-bit touched mask
//   S2  wave w takes the touched supers with ordinal = w mod 4, tests their 32 buckets, appends the
//       touched buckets to an LDS work list                                      -> barrier
//   S3  waves update touched buckets, three per wave in flight (loads batched)   -> barrier
//   S4  waves recompute the key of the supers they own in this step              -> barrier
//   S5  every wave (redundantly): arg-max over the <= 64 super keys, winner coordinates from LDS
// prev_idx/prev_offset (optional): samples already computed on this workspace by an earlier call for the
// same cloud (FPS is deterministic: a shorter request is a prefix of a longer one) — copied, then resumed.
constexpr int FPS_SUPER = 32;                                 // buckets per super-bucket
constexpr int FPS_MAX_SUPERS = FPS_MAX_BUCKETS / FPS_SUPER;   // 64
constexpr int FPS_NW = 4;

struct KeyMax {
    unsigned long long key;  // wave maximum
    int lane;                // a lane holding it (unique when keys are unique)
};
// 64-bit wave max: DPP max of the high words; the low words only need a second pass when several lanes
// share the maximal high word (exact distance ties).
__device__ __forceinline__ KeyMax wave_key_max(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = wave_max_u32(hi);
    const unsigned long long tied = __ballot(hi == mh);
    KeyMax r;
    if (__popcll(tied) == 1) {
        r.lane = __ffsll(tied) - 1;
        r.key = ((unsigned long long)mh << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, r.lane);
    } else {
        const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
        r.key = ((unsigned long long)mh << 32) | ml;
        r.lane = __ffsll((unsigned long long)__ballot(hi == mh && lo == ml)) - 1;
    }
    return r;
}

__global__ __launch_bounds__(FPS_NW * 64) void fps_bucket_kernel(int Bref, int log2B, int BSZ, const float *__restrict__ xyz,
                                                                 const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                 float4 *__restrict__ pts, const unsigned *__restrict__ rank,
                                                                 const int *__restrict__ prev_idx, const int *__restrict__ prev_offset,
                                                                 int *__restrict__ idx) {
    constexpr int NT = FPS_NW * 64;
    extern __shared__ unsigned long long smem64[];
    unsigned long long *bkey = smem64;                                  // [MAXB]
    float *bminx = reinterpret_cast<float *>(bkey + FPS_MAX_BUCKETS);  // 6 x [MAXB] box, 3 x [MAXB] arg-max coords
    float *bminy = bminx + FPS_MAX_BUCKETS, *bminz = bminy + FPS_MAX_BUCKETS;
    float *bmaxx = bminz + FPS_MAX_BUCKETS, *bmaxy = bmaxx + FPS_MAX_BUCKETS, *bmaxz = bmaxy + FPS_MAX_BUCKETS;
    float *bestx = bmaxz + FPS_MAX_BUCKETS, *besty = bestx + FPS_MAX_BUCKETS, *bestz = besty + FPS_MAX_BUCKETS;
    int *worklist = reinterpret_cast<int *>(bestz + FPS_MAX_BUCKETS);  // [MAXB]
    __shared__ unsigned long long skey[FPS_MAX_SUPERS];
    __shared__ float sbox[6][FPS_MAX_SUPERS];
    __shared__ float sbest[3][FPS_MAX_SUPERS];
    __shared__ int wl_count;

    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    if (end_n <= start_n) {
        for (int j = start_m + tid; j < end_m; j += NT) idx[j] = start_n;
        return;
    }
    const int n = end_n - start_n;
    const int nb = (n + BSZ - 1) / BSZ;
    const int ns = (nb + FPS_SUPER - 1) / FPS_SUPER;

    // samples inherited from the previous call on this state
    int done = 0;
    if (prev_idx) {
        const int ps = bid == 0 ? 0 : prev_offset[bid - 1], pe = prev_offset[bid];
        done = min(pe - ps, end_m - start_m);
        for (int t = tid; t < done; t += NT) idx[start_m + t] = prev_idx[ps + t];
    }
    if (start_m + done >= end_m) return;

    // bucket boxes and cached keys from the current min-dist field (1e10 everywhere on a fresh start,
    // so the first step touches every bucket)
    for (int bk = wave; bk < nb; bk += FPS_NW) {
        const int p0 = start_n + bk * BSZ, p1 = min(p0 + BSZ, end_n);
        float mnx = INFINITY, mny = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxy = -INFINITY, mxz = -INFINITY;
        unsigned long long best = 0ull;
        float bx = 0.f, by = 0.f, bz = 0.f;
        for (int pos = p0 + lane; pos < p1; pos += 64) {
            const float4 p = pts[pos];
            mnx = fminf(mnx, p.x); mny = fminf(mny, p.y); mnz = fminf(mnz, p.z);
            mxx = fmaxf(mxx, p.x); mxy = fmaxf(mxy, p.y); mxz = fmaxf(mxz, p.z);
            const unsigned long long key = ((unsigned long long)__float_as_uint(p.w) << 32) | rank[pos];
            if (key > best) { best = key; bx = p.x; by = p.y; bz = p.z; }
        }
        for (int st = 1; st < 64; st <<= 1) {
            mnx = fminf(mnx, __shfl_xor(mnx, st, 64)); mny = fminf(mny, __shfl_xor(mny, st, 64)); mnz = fminf(mnz, __shfl_xor(mnz, st, 64));
            mxx = fmaxf(mxx, __shfl_xor(mxx, st, 64)); mxy = fmaxf(mxy, __shfl_xor(mxy, st, 64)); mxz = fmaxf(mxz, __shfl_xor(mxz, st, 64));
        }
        const KeyMax km = wave_key_max(best);
        if (lane == 0) {
            bminx[bk] = mnx; bminy[bk] = mny; bminz[bk] = mnz;
            bmaxx[bk] = mxx; bmaxy[bk] = mxy; bmaxz[bk] = mxz;
        }
        if (lane == km.lane) {
            bkey[bk] = km.key;
            bestx[bk] = bx; besty[bk] = by; bestz[bk] = bz;
        }
    }
    if (tid == 0) {
        wl_count = 0;
        if (done == 0) idx[start_m] = start_n;
    }
    __syncthreads();  // bucket records complete; also orders the idx[] copy above before the read below
    // super-bucket boxes and keys
    auto refresh_super = [&](int sb) {  // whole wave; lanes 0..31 <-> the super's buckets
        const int bk = sb * FPS_SUPER + lane;
        const bool ok = lane < FPS_SUPER && bk < nb;
        const KeyMax km = wave_key_max(ok ? bkey[bk] : 0ull);
        if (lane == km.lane) {
            skey[sb] = km.key;
            sbest[0][sb] = bestx[bk]; sbest[1][sb] = besty[bk]; sbest[2][sb] = bestz[bk];
        }
    };
    for (int sb = wave; sb < ns; sb += FPS_NW) {
        const int bk = sb * FPS_SUPER + lane;
        const bool ok = lane < FPS_SUPER && bk < nb;
        float v[6] = {ok ? bminx[bk] : INFINITY, ok ? bminy[bk] : INFINITY, ok ? bminz[bk] : INFINITY,
                      ok ? bmaxx[bk] : -INFINITY, ok ? bmaxy[bk] : -INFINITY, ok ? bmaxz[bk] : -INFINITY};
        for (int st = 1; st < 64; st <<= 1)
            for (int a = 0; a < 6; a++) {
                const float o = __shfl_xor(v[a], st, 64);
                v[a] = a < 3 ? fminf(v[a], o) : fmaxf(v[a], o);
            }
        if (lane < 6) sbox[lane][sb] = v[lane];
        refresh_super(sb);
    }
    __syncthreads();
    const int first = done == 0 ? start_n : idx[start_m + done - 1];
    float x1 = xyz[(size_t)first * 3 + 0], y1 = xyz[(size_t)first * 3 + 1], z1 = xyz[(size_t)first * 3 + 2];

    for (int j = start_m + max(done, 1); j < end_m; j++) {
        // S1: touched supers (same result in every wave)
        bool st_touched = false;
        if (lane < ns) {
            const float dx = fmaxf(fmaxf(sbox[0][lane] - x1, x1 - sbox[3][lane]), 0.f);
            const float dy = fmaxf(fmaxf(sbox[1][lane] - y1, y1 - sbox[4][lane]), 0.f);
            const float dz = fmaxf(fmaxf(sbox[2][lane] - z1, z1 - sbox[5][lane]), 0.f);
            st_touched = sqd(dx, dy, dz) < __uint_as_float((unsigned)(skey[lane] >> 32));
        }
        const unsigned long long smask = __ballot(st_touched);
        // S2: buckets of the supers this wave owns in this step
        {
            unsigned long long rest = smask;
            int ord = 0;
            while (rest) {
                const int sb = __ffsll(rest) - 1;
                rest &= rest - 1;
                if ((ord++ & (FPS_NW - 1)) != wave) continue;
                const int bk = sb * FPS_SUPER + lane;
                bool hit = false;
                if (lane < FPS_SUPER && bk < nb) {
                    const float dx = fmaxf(fmaxf(bminx[bk] - x1, x1 - bmaxx[bk]), 0.f);
                    const float dy = fmaxf(fmaxf(bminy[bk] - y1, y1 - bmaxy[bk]), 0.f);
                    const float dz = fmaxf(fmaxf(bminz[bk] - z1, z1 - bmaxz[bk]), 0.f);
                    hit = sqd(dx, dy, dz) < __uint_as_float((unsigned)(bkey[bk] >> 32));
                }
                const unsigned long long hm = __ballot(hit);
                if (hm) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&wl_count, __popcll(hm));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (hit) worklist[base + __popcll(hm & ((1ull << lane) - 1))] = bk;
                }
            }
        }
        __syncthreads();
        const int cnt = wl_count;
        // S3: update touched buckets; three buckets per wave in flight when a bucket is one wave wide
        if (BSZ == 64) {
            for (int w0 = wave * 3; w0 < cnt; w0 += FPS_NW * 3) {
                int bk[3];
                float4 p[3];
                unsigned rk[3];
                bool ok[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    bk[u] = w0 + u < cnt ? worklist[w0 + u] : -1;
                    const int pos = start_n + bk[u] * 64 + lane;
                    ok[u] = bk[u] >= 0 && pos < end_n;
                    if (ok[u]) { p[u] = pts[pos]; rk[u] = rank[pos]; }
                }
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    if (bk[u] < 0) continue;  // wave-uniform
                    unsigned long long key = 0ull;
                    if (ok[u]) {
                        const float d = sqd(p[u].x - x1, p[u].y - y1, p[u].z - z1);
                        const float d2 = fminf(d, p[u].w);
                        if (d2 != p[u].w) reinterpret_cast<float *>(pts + start_n + bk[u] * 64 + lane)[3] = d2;
                        key = ((unsigned long long)__float_as_uint(d2) << 32) | rk[u];
                    }
                    const KeyMax km = wave_key_max(key);
                    if (lane == km.lane) {
                        bkey[bk[u]] = km.key;
                        bestx[bk[u]] = p[u].x; besty[bk[u]] = p[u].y; bestz[bk[u]] = p[u].z;
                    }
                }
            }
        } else {
            for (int w = wave; w < cnt; w += FPS_NW) {
                const int bk = worklist[w];
                const int p0 = start_n + bk * BSZ, p1 = min(p0 + BSZ, end_n);
                unsigned long long best = 0ull;
                float bx = 0.f, by = 0.f, bz = 0.f;
                for (int pos = p0 + lane; pos < p1; pos += 64) {
                    const float4 p = pts[pos];
                    const unsigned rk = rank[pos];
                    const float d = sqd(p.x - x1, p.y - y1, p.z - z1);
                    const float d2 = fminf(d, p.w);
                    if (d2 != p.w) reinterpret_cast<float *>(pts + pos)[3] = d2;
                    const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | rk;
                    if (key > best) { best = key; bx = p.x; by = p.y; bz = p.z; }
                }
                const KeyMax km = wave_key_max(best);
                if (lane == km.lane) {
                    bkey[bk] = km.key;
                    bestx[bk] = bx; besty[bk] = by; bestz[bk] = bz;
                }
            }
        }
        __syncthreads();
        // S4: keys of the touched supers (same ownership as S2)
        {
            unsigned long long rest = smask;
            int ord = 0;
            while (rest) {
                const int sb = __ffsll(rest) - 1;
                rest &= rest - 1;
                if ((ord++ & (FPS_NW - 1)) != wave) continue;
                refresh_super(sb);
            }
            if (tid == 0) wl_count = 0;
        }
        __syncthreads();
        // S5: arg-max over super keys (every wave computes the same winner)
        const KeyMax km = wave_key_max(lane < ns ? skey[lane] : 0ull);
        x1 = sbest[0][km.lane]; y1 = sbest[1][km.lane]; z1 = sbest[2][km.lane];
        if (tid == 0) idx[j] = start_n + rel_of(km.key, Bref, log2B);
    }
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static int bits_for(int b) {
    int r = 1;
    while ((1 << r) < b) r++;
    return r;
}
static size_t fps_cub_bytes(int b, int N) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (const int *)nullptr, (int *)nullptr, N, 0, 32 + bits_for(b), (hipStream_t) nullptr);
    return bytes;
}

struct FpsResume {
    const int *prev_idx = nullptr;
    const int *prev_offset = nullptr;
};
FpsResume &fps_resume() {
    static thread_local FpsResume r;
    return r;
}

// returns false when the bucket path does not apply (caller falls back to the block kernel)
bool fps_bucket_launch(int b, int n, int Bref, int log2B, const float *xyz, const int *offset, const int *new_offset,
                       int N_total, int *idx) {
    Workspace &w = workspace();
    if (w.ptr == nullptr || N_total <= 0) return false;
    const size_t need = pointops2_fps_workspace_bytes(b, N_total);
    if (w.bytes < need) return false;
    hipStream_t st = state().stream;
    char *p = reinterpret_cast<char *>(w.ptr);
    const size_t f4 = al((size_t)N_total * 4), f8 = al((size_t)N_total * 8), f16 = al((size_t)N_total * 16);
    // persistent part of the workspace (what a resumed call needs)
    float4 *pts = (float4 *)p; p += f16;
    unsigned *rank = (unsigned *)p; p += f4;
    // set-up scratch
    int *sorig = (int *)p; p += f4;
    int *vals_in = (int *)p; p += f4;
    unsigned long long *keys_in = (unsigned long long *)p; p += f8;
    unsigned long long *keys_out = (unsigned long long *)p; p += f8;
    float *bbox = (float *)p; p += al((size_t)b * 6 * 4);
    void *cub_tmp = p;
    size_t cub_bytes = w.bytes - (size_t)(p - reinterpret_cast<char *>(w.ptr));
    FpsResume rs = fps_resume();
    fps_resume() = FpsResume();
    if (rs.prev_idx == nullptr) {
        hipLaunchKernelGGL(fps_bbox_kernel, dim3(b), dim3(256), 0, st, xyz, offset, bbox);
        hipLaunchKernelGGL(fps_morton_kernel, dim3(div_up(N_total, 256)), dim3(256), 0, st, N_total, b, xyz, offset, bbox, keys_in, vals_in);
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys_in, keys_out, (const int *)vals_in, sorig, N_total, 0,
                                                          32 + bits_for(b), st);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return true; }
        hipLaunchKernelGGL(fps_gather_kernel, dim3(div_up(N_total, 256)), dim3(256), 0, st, N_total, b, Bref, log2B, xyz, offset, sorig, pts, rank);
    }
    const int BSZ = 64 * div_up(n, 64 * FPS_MAX_BUCKETS);
    const size_t lds = (size_t)FPS_MAX_BUCKETS * (8 + 9 * 4 + 4);
    allow_big_lds(fps_bucket_kernel, lds);
    hipLaunchKernelGGL(fps_bucket_kernel, dim3(b), dim3(FPS_NW * 64), lds, st, Bref, log2B, BSZ, xyz, offset, new_offset, pts, rank,
                       rs.prev_idx, rs.prev_offset, idx);
    return true;
}

}  // namespace p2

using namespace p2;

extern "C" {

void pointops2_set_workspace(void *ptr, size_t bytes) {
    workspace().ptr = ptr;
    workspace().bytes = bytes;
}

size_t pointops2_fps_workspace_bytes(int b, int N) {
    if (b <= 0 || N <= 0) return 0;
    return al((size_t)N * 16) + 3 * al((size_t)N * 4) + 2 * al((size_t)N * 8) + al((size_t)b * 6 * 4) + al(fps_cub_bytes(b, N));
}

void pointops2_set_fps_resume(const int *prev_idx, const int *prev_offset) {
    fps_resume().prev_idx = prev_idx;
    fps_resume().prev_offset = prev_offset;
}

}  // extern "C"
