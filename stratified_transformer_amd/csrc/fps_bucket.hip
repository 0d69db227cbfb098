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
// Static ownership, one barrier per step.  Bucket b belongs to wave b % NW, and inside that wave to one
// lane, which keeps the bucket's box, cached key and arg-max coordinates IN REGISTERS (NBL buckets per
// lane).  A step, per wave, with no synchronisation:
//     test own buckets against the new sample (registers only)
//     update the touched ones: 64 lanes <-> 64 points, up to four buckets' loads in flight; only the owning
//       wave ever loads or stores a bucket's points, so program order alone keeps the min-dist field coherent
//       and the step barrier does not have to wait for the stores to be acknowledged
//     reduce the wave's best key -> LDS slot (ping-pong by step parity)
// then ONE raw s_barrier (LDS only), and every wave reduces the NW slots to the winner and its coordinates.
// prev_idx/prev_offset (optional): samples already computed on this workspace by an earlier call for the
// same cloud (FPS is deterministic: a shorter request is a prefix of a longer one) — copied, then resumed.
constexpr int FPS_NW = 8;

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

__device__ __forceinline__ void lds_barrier() {
    // LDS traffic of this wave retired, then the workgroup barrier; global stores stay in flight
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int NBL>
__global__ __launch_bounds__(FPS_NW * 64) void fps_bucket_kernel(int Bref, int log2B, int BSZ, const float *__restrict__ xyz,
                                                                 const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                 float4 *__restrict__ pts, const unsigned *__restrict__ rank,
                                                                 const int *__restrict__ prev_idx, const int *__restrict__ prev_offset,
                                                                 int *__restrict__ idx) {
    constexpr int NT = FPS_NW * 64;
    __shared__ unsigned long long wkey[2][FPS_NW];
    __shared__ float wbest[2][FPS_NW][4];

    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    if (end_n <= start_n) {
        for (int j = start_m + tid; j < end_m; j += NT) idx[j] = start_n;
        return;
    }
    const int n = end_n - start_n;
    const int nb = (n + BSZ - 1) / BSZ;

    // samples inherited from the previous call on this state
    int done = 0;
    if (prev_idx) {
        const int ps = bid == 0 ? 0 : prev_offset[bid - 1], pe = prev_offset[bid];
        done = min(pe - ps, end_m - start_m);
        for (int t = tid; t < done; t += NT) idx[start_m + t] = prev_idx[ps + t];
    }
    if (start_m + done >= end_m) return;

    // owned buckets: slot s of lane l  <->  bucket (s*64 + l)*NW + wave
    float mnx[NBL], mny[NBL], mnz[NBL], mxx[NBL], mxy[NBL], mxz[NBL], bx[NBL], by[NBL], bz[NBL];
    unsigned long long key[NBL];
#pragma unroll
    for (int s = 0; s < NBL; s++) {
        mnx[s] = mny[s] = mnz[s] = INFINITY;       // an absent bucket never passes the test below
        mxx[s] = mxy[s] = mxz[s] = -INFINITY;
        bx[s] = by[s] = bz[s] = 0.f;
        key[s] = 0ull;
    }
    // box + cached key of every owned bucket from the current min-dist field (1e10 everywhere on a fresh
    // start, so the first step touches every bucket)
#pragma unroll
    for (int s = 0; s < NBL; s++) {
        for (int l = 0; l < 64; l++) {
            const int bk = (s * 64 + l) * FPS_NW + wave;
            if (bk >= nb) break;
            const int p0 = start_n + bk * BSZ, p1 = min(p0 + BSZ, end_n);
            float a0 = INFINITY, a1 = INFINITY, a2 = INFINITY, b0 = -INFINITY, b1 = -INFINITY, b2 = -INFINITY;
            unsigned long long best = 0ull;
            float cx = 0.f, cy = 0.f, cz = 0.f;
            for (int pos = p0 + lane; pos < p1; pos += 64) {
                const float4 p = pts[pos];
                a0 = fminf(a0, p.x); a1 = fminf(a1, p.y); a2 = fminf(a2, p.z);
                b0 = fmaxf(b0, p.x); b1 = fmaxf(b1, p.y); b2 = fmaxf(b2, p.z);
                const unsigned long long k = ((unsigned long long)__float_as_uint(p.w) << 32) | rank[pos];
                if (k > best) { best = k; cx = p.x; cy = p.y; cz = p.z; }
            }
            for (int st = 1; st < 64; st <<= 1) {
                a0 = fminf(a0, __shfl_xor(a0, st, 64)); a1 = fminf(a1, __shfl_xor(a1, st, 64)); a2 = fminf(a2, __shfl_xor(a2, st, 64));
                b0 = fmaxf(b0, __shfl_xor(b0, st, 64)); b1 = fmaxf(b1, __shfl_xor(b1, st, 64)); b2 = fmaxf(b2, __shfl_xor(b2, st, 64));
            }
            const KeyMax km = wave_key_max(best);
            const float wx = __shfl(cx, km.lane, 64), wy = __shfl(cy, km.lane, 64), wz = __shfl(cz, km.lane, 64);
            if (lane == l) {
                mnx[s] = a0; mny[s] = a1; mnz[s] = a2; mxx[s] = b0; mxy[s] = b1; mxz[s] = b2;
                key[s] = km.key; bx[s] = wx; by[s] = wy; bz[s] = wz;
            }
        }
    }
    if (tid == 0 && done == 0) idx[start_m] = start_n;
    __syncthreads();  // orders the idx[] copy above before the read below
    const int first = done == 0 ? start_n : idx[start_m + done - 1];
    float x1 = xyz[(size_t)first * 3 + 0], y1 = xyz[(size_t)first * 3 + 1], z1 = xyz[(size_t)first * 3 + 2];

    int par = 0;
    for (int j = start_m + max(done, 1); j < end_m; j++, par ^= 1) {
#pragma unroll
        for (int s = 0; s < NBL; s++) {
            const float dx = fmaxf(fmaxf(mnx[s] - x1, x1 - mxx[s]), 0.f);
            const float dy = fmaxf(fmaxf(mny[s] - y1, y1 - mxy[s]), 0.f);
            const float dz = fmaxf(fmaxf(mnz[s] - z1, z1 - mxz[s]), 0.f);
            const bool hit = sqd(dx, dy, dz) < __uint_as_float((unsigned)(key[s] >> 32));
            unsigned long long hm = __ballot(hit);
            while (hm) {
                // up to four touched buckets of this slot: issue all their loads, then reduce one by one
                int ln[4];
                float4 p[4];
                unsigned rk[4];
                bool ok[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    ln[u] = hm ? __ffsll(hm) - 1 : -1;
                    if (hm) hm &= hm - 1;
                }
                if (BSZ == 64) {
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int pos = start_n + ((s * 64 + ln[u]) * FPS_NW + wave) * 64 + lane;
                        ok[u] = ln[u] >= 0 && pos < end_n;
                        if (ok[u]) { p[u] = pts[pos]; rk[u] = rank[pos]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (ln[u] < 0) continue;  // wave-uniform
                    unsigned long long best = 0ull;
                    float cx = 0.f, cy = 0.f, cz = 0.f;
                    if (BSZ == 64) {
                        if (ok[u]) {
                            const float d = sqd(p[u].x - x1, p[u].y - y1, p[u].z - z1);
                            const float d2 = fminf(d, p[u].w);
                            if (d2 != p[u].w)
                                reinterpret_cast<float *>(pts + start_n + ((s * 64 + ln[u]) * FPS_NW + wave) * 64 + lane)[3] = d2;
                            best = ((unsigned long long)__float_as_uint(d2) << 32) | rk[u];
                            cx = p[u].x; cy = p[u].y; cz = p[u].z;
                        }
                    } else {
                        const int bk = (s * 64 + ln[u]) * FPS_NW + wave;
                        const int p0 = start_n + bk * BSZ, p1 = min(p0 + BSZ, end_n);
                        for (int pos = p0 + lane; pos < p1; pos += 64) {
                            const float4 q = pts[pos];
                            const float d = sqd(q.x - x1, q.y - y1, q.z - z1);
                            const float d2 = fminf(d, q.w);
                            if (d2 != q.w) reinterpret_cast<float *>(pts + pos)[3] = d2;
                            const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | rank[pos];
                            if (k > best) { best = k; cx = q.x; cy = q.y; cz = q.z; }
                        }
                    }
                    const KeyMax km = wave_key_max(best);
                    const float wx = __shfl(cx, km.lane, 64), wy = __shfl(cy, km.lane, 64), wz = __shfl(cz, km.lane, 64);
                    if (lane == ln[u]) { key[s] = km.key; bx[s] = wx; by[s] = wy; bz[s] = wz; }
                }
            }
        }
        // the wave's best bucket
        unsigned long long mk = key[0];
        float mx_ = bx[0], my_ = by[0], mz_ = bz[0];
#pragma unroll
        for (int s = 1; s < NBL; s++)
            if (key[s] > mk) { mk = key[s]; mx_ = bx[s]; my_ = by[s]; mz_ = bz[s]; }
        const KeyMax wm = wave_key_max(mk);
        if (lane == wm.lane) {
            wkey[par][wave] = wm.key;
            wbest[par][wave][0] = mx_; wbest[par][wave][1] = my_; wbest[par][wave][2] = mz_;
        }
        lds_barrier();
        const KeyMax gm = wave_key_max(lane < FPS_NW ? wkey[par][lane] : 0ull);
        x1 = wbest[par][gm.lane][0]; y1 = wbest[par][gm.lane][1]; z1 = wbest[par][gm.lane][2];
        if (tid == 0) idx[j] = start_n + rel_of(gm.key, Bref, log2B);
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
    const int nbuckets = div_up(n, BSZ);
    const int per_lane = div_up(nbuckets, FPS_NW * 64);
    if (per_lane <= 1)
        hipLaunchKernelGGL(fps_bucket_kernel<1>, dim3(b), dim3(FPS_NW * 64), 0, st, Bref, log2B, BSZ, xyz, offset, new_offset, pts, rank,
                           rs.prev_idx, rs.prev_offset, idx);
    else if (per_lane <= 2)
        hipLaunchKernelGGL(fps_bucket_kernel<2>, dim3(b), dim3(FPS_NW * 64), 0, st, Bref, log2B, BSZ, xyz, offset, new_offset, pts, rank,
                           rs.prev_idx, rs.prev_offset, idx);
    else
        hipLaunchKernelGGL(fps_bucket_kernel<4>, dim3(b), dim3(FPS_NW * 64), 0, st, Bref, log2B, BSZ, xyz, offset, new_offset, pts, rank,
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
