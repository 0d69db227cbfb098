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
#include "fps_common.h"
#include <hipcub/hipcub.hpp>
#include <cstdio>
#include <cstdlib>

namespace p2 {

Workspace &workspace() {
    static thread_local Workspace w;
    return w;
}

constexpr int FPS_MAX_BUCKETS = 2048;

// maximum over lanes 0..15 only (values in the first row): four row steps, result in every lane of row 0
__device__ __forceinline__ unsigned row0_max_u32(unsigned v) {
    v = dpp_step<0xB1, 0xF>(v);
    v = dpp_step<0x4E, 0xF>(v);
    v = dpp_step<0x141, 0xF>(v);
    v = dpp_step<0x140, 0xF>(v);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
// max of (hi:lo) keys: max hi first, then max lo among the lanes that hold it
__device__ __forceinline__ unsigned long long wmax64(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = wave_max_u32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    return ((unsigned long long)mh << 32) | ml;
}

// ---- set-up ----------------------------------------------------------------------------------
// one workgroup of 1024 threads per cloud, four points in flight per thread (256 threads, one point at a time: 95 us for a
// 100 000-point cloud, in front of the sampler on the pass's start-up path)
constexpr int BBOX_T = 1024;
__global__ __launch_bounds__(BBOX_T) void fps_bbox_kernel(const float *__restrict__ xyz, const int *__restrict__ offset,
                                                          float *__restrict__ bbox) {
    __shared__ float red[6][BBOX_T / 64];
    const int bid = blockIdx.x, tid = threadIdx.x;
    const int s = bid == 0 ? 0 : offset[bid - 1], e = offset[bid];
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i0 = s + tid; i0 < e; i0 += 4 * BBOX_T) {
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = min(i0 + u * BBOX_T, e - 1);  // (past the end: the last point again)
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
        float r = red[tid][0];
        for (int w = 1; w < BBOX_T / 64; w++) r = tid < 3 ? fminf(r, red[tid][w]) : fmaxf(r, red[tid][w]);
        bbox[bid * 6 + tid] = r;
    }
}

void launch_bbox(int b, const float *xyz, const int *offset, float *bbox, hipStream_t st) {
    hipLaunchKernelGGL(fps_bbox_kernel, dim3(b), dim3(BBOX_T), 0, st, xyz, offset, bbox);
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
                                  const int *__restrict__ order, float4 *__restrict__ pts, unsigned *__restrict__ rank,
                                  int *__restrict__ inv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int o = order[i];
    inv[o] = i;
    pts[i] = make_float4(xyz[(size_t)o * 3 + 0], xyz[(size_t)o * 3 + 1], xyz[(size_t)o * 3 + 2], 1e10f);  // pointops.py:26
    int bid = 0;
    while (bid < b - 1 && i >= offset[bid]) bid++;
    const int start_n = bid == 0 ? 0 : offset[bid - 1];
    rank[i] = (unsigned)key_of(0.f, o - start_n, Bref, log2B);
}

// ---- identity-prefix verification ----------------------------------------------------------------
// A cloud that is itself the output of an earlier FPS (every stage after the first: TransitionDown keeps
// the samples in selection order, stratified_transformer.py:103-104) is sampled again as 0, 1, 2, ...:
// the arg-max property of a prefix is inherited by any subset that contains it.  Exact ties aside, the
// m dependent steps collapse into a CHECK that is embarrassingly parallel: sample j is right iff no point
// has a larger key than point j after the first j samples.  The check is exact (same fma chain, same tie
// ranks); where it fails (first step of an unordered cloud, or a genuine tie) the sequential sampler
// takes over from the verified prefix.
constexpr int VER_T = 256;

// thresholds: T[j] = key of point j after samples 0..j-1, for j in [jlo, jhi).  A workgroup = 64 points x 4 waves: wave sl takes
// the sl-th quarter of every 256-sample tile, the four partial minima meet in LDS (one point per thread left the chip three
// quarters empty at 25 000 points: 98 workgroups, every thread a chain of 6 000 dependent steps).
constexpr int VER_P = 64;  // points per workgroup of the threshold / rebuild kernels
__global__ __launch_bounds__(VER_T) void fps_verify_threshold_kernel(int Bref, int log2B, int jlo, const float *__restrict__ xyz,
                                                                     const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                     const int *__restrict__ first_bad, unsigned long long *__restrict__ T) {
    const int bid = blockIdx.y;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    const int m = min(end_m - start_m, end_n - start_n);  // the identity prefix cannot be longer than the cloud
    if (first_bad[bid] < jlo) return;
    const int p = threadIdx.x & (VER_P - 1), sl = threadIdx.x / VER_P;
    const int j = jlo + blockIdx.x * VER_P + p;
    __shared__ float4 s4[VER_T];
    __shared__ float smin[VER_T / VER_P][VER_P];
    const int jmax = min(jlo + (int)(blockIdx.x + 1) * VER_P, m);  // tiles of samples needed by this block: i < jmax
    if (jlo + (int)blockIdx.x * VER_P >= m) return;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (j < m) { px = xyz[(size_t)(start_n + j) * 3]; py = xyz[(size_t)(start_n + j) * 3 + 1]; pz = xyz[(size_t)(start_n + j) * 3 + 2]; }
    float D = 1e10f;
    for (int i0 = 0; i0 < jmax; i0 += VER_T) {
        __syncthreads();
        const int i = i0 + threadIdx.x;
        if (i < jmax) s4[threadIdx.x] = make_float4(xyz[(size_t)(start_n + i) * 3], xyz[(size_t)(start_n + i) * 3 + 1], xyz[(size_t)(start_n + i) * 3 + 2], 0.f);
        __syncthreads();
        const int lim = min(VER_T, j - i0);  // samples i < j
        const int t1 = min(lim, (sl + 1) * VER_P);
#pragma unroll 8
        for (int t = sl * VER_P; t < t1; t++) {
            const float4 sp = s4[t];
            D = fminf(D, sqd(px - sp.x, py - sp.y, pz - sp.z));
        }
    }
    smin[sl][p] = D;
    __syncthreads();
    if (sl == 0 && j < m) {
#pragma unroll
        for (int w = 1; w < VER_T / VER_P; w++) D = fminf(D, smin[w][p]);
        T[start_n + j] = ((unsigned long long)__float_as_uint(D) << 32) | (unsigned)key_of(0.f, j, Bref, log2B);
    }
}

// scan: every point x checks key_x(j) <= T[j] for j in [jlo, jhi); first violation -> first_bad (atomic min).
// 64 points x 4 waves per workgroup like the threshold kernel, but the check of step j needs the minimum over ALL samples before j:
// per 256-sample tile, wave sl first takes the minimum over its quarter (no checks), the quarters meet in LDS, then it walks its
// quarter again from min(everything before the tile, the earlier quarters) with the checks - twice the distance evaluations for
// four times the workgroups.
__global__ __launch_bounds__(VER_T) void fps_verify_scan_kernel(int Bref, int log2B, int jlo, int jhi_cap, const float *__restrict__ xyz,
                                                                const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                                const unsigned long long *__restrict__ T, int *__restrict__ first_bad) {
    const int bid = blockIdx.y;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    const int n = end_n - start_n;
    const int m = min(min(end_m - start_m, n), jhi_cap);
    if (first_bad[bid] < jlo || (int)blockIdx.x * VER_P >= n) return;
    const int p = threadIdx.x & (VER_P - 1), sl = threadIdx.x / VER_P;
    const int x = blockIdx.x * VER_P + p;
    __shared__ float4 s4[VER_T];
    __shared__ unsigned long long sT[VER_T];
    __shared__ float smin[VER_T / VER_P][VER_P];
    __shared__ int s_fb;
    float px = 0.f, py = 0.f, pz = 0.f;
    const bool live = x < n;
    if (live) { px = xyz[(size_t)(start_n + x) * 3]; py = xyz[(size_t)(start_n + x) * 3 + 1]; pz = xyz[(size_t)(start_n + x) * 3 + 2]; }
    const unsigned rk = (unsigned)key_of(0.f, x, Bref, log2B);
    float Dprev = 1e10f;  // minimum over the samples of all earlier tiles
    int bad = 0x7fffffff;
    // step j uses samples 0..j-1: tile t holds samples i0..i0+255 and thresholds of steps i0+1..i0+256
    for (int i0 = 0; i0 < m - 1; i0 += VER_T) {
        __syncthreads();
        if (threadIdx.x == 0) s_fb = *reinterpret_cast<volatile int *>(first_bad + bid);
        __syncthreads();
        if (i0 > s_fb) break;  // block-uniform: an earlier step already failed somewhere
        const int i = i0 + threadIdx.x;
        if (i < m - 1) {
            s4[threadIdx.x] = make_float4(xyz[(size_t)(start_n + i) * 3], xyz[(size_t)(start_n + i) * 3 + 1], xyz[(size_t)(start_n + i) * 3 + 2], 0.f);
            sT[threadIdx.x] = (i + 1 >= jlo) ? T[start_n + i + 1] : ~0ull;
        } else {
            s4[threadIdx.x] = make_float4(1e18f, 1e18f, 1e18f, 0.f);  // (a padded sample is infinitely far away and its threshold never violated)
            sT[threadIdx.x] = ~0ull;
        }
        __syncthreads();
        const int t0 = sl * VER_P;
        float mine = 1e10f;  // this quarter's minimum
#pragma unroll 8
        for (int u = 0; u < VER_P; u++) {
            const float4 sp = s4[t0 + u];
            mine = fminf(mine, sqd(px - sp.x, py - sp.y, pz - sp.z));
        }
        smin[sl][p] = mine;
        __syncthreads();
        float D = Dprev;
#pragma unroll
        for (int w = 0; w < VER_T / VER_P; w++) {
            const float o = smin[w][p];
            D = w < sl ? fminf(D, o) : D;
            Dprev = fminf(Dprev, o);
        }
        if (live && bad == 0x7fffffff) {
            // no early exit inside a quarter: eight samples per trip, the LDS reads of a trip are independent of its compares
            for (int u0 = 0; u0 < VER_P; u0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int t = t0 + u0 + u;
                    const float4 sp = s4[t];
                    D = fminf(D, sqd(px - sp.x, py - sp.y, pz - sp.z));
                    const unsigned long long key = ((unsigned long long)__float_as_uint(D) << 32) | rk;
                    if (key > sT[t] && bad == 0x7fffffff) bad = i0 + t + 1;
                }
            }
        }
    }
    if (bad != 0x7fffffff) atomicMin(first_bad + bid, bad);
}

__global__ void fps_verify_init_kernel(int b, const int *__restrict__ offset, const int *__restrict__ new_offset, int *__restrict__ first_bad) {
    const int bid = blockIdx.x * blockDim.x + threadIdx.x;
    if (bid >= b) return;
    const int n = offset[bid] - (bid == 0 ? 0 : offset[bid - 1]);
    const int m = new_offset[bid] - (bid == 0 ? 0 : new_offset[bid - 1]);
    first_bad[bid] = max(min(m, n), 0);  // = number of samples that are verified to be 0,1,2,... (lowered by the scans)
}

// min-dist field after the verified prefix (samples 0..v-2 applied), written in sampler (Morton) order;
// only where the verified prefix is longer than what the kept sampler state already covers
__global__ __launch_bounds__(VER_T) void fps_rebuild_kernel(const float *__restrict__ xyz, const int *__restrict__ offset,
                                                            const int *__restrict__ prev_offset, const int *__restrict__ first_bad,
                                                            const int *__restrict__ inv, float4 *__restrict__ pts) {
    const int bid = blockIdx.y;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int n = end_n - start_n;
    const int prev = prev_offset ? prev_offset[bid] - (bid == 0 ? 0 : prev_offset[bid - 1]) : 0;
    const int v = first_bad[bid];
    if (v <= prev || v <= 1 || (int)blockIdx.x * VER_P >= n) return;
    const int p = threadIdx.x & (VER_P - 1), sl = threadIdx.x / VER_P;  // 64 points x 4 quarter-tiles (see the threshold kernel)
    const int x = blockIdx.x * VER_P + p;
    __shared__ float4 s4[VER_T];
    __shared__ float smin[VER_T / VER_P][VER_P];
    float px = 0.f, py = 0.f, pz = 0.f;
    if (x < n) { px = xyz[(size_t)(start_n + x) * 3]; py = xyz[(size_t)(start_n + x) * 3 + 1]; pz = xyz[(size_t)(start_n + x) * 3 + 2]; }
    float D = 1e10f;
    for (int i0 = 0; i0 < v - 1; i0 += VER_T) {
        __syncthreads();
        const int i = i0 + threadIdx.x;
        if (i < v - 1) s4[threadIdx.x] = make_float4(xyz[(size_t)(start_n + i) * 3], xyz[(size_t)(start_n + i) * 3 + 1], xyz[(size_t)(start_n + i) * 3 + 2], 0.f);
        __syncthreads();
        const int lim = min(VER_T, v - 1 - i0);
        const int t1 = min(lim, (sl + 1) * VER_P);
#pragma unroll 8
        for (int t = sl * VER_P; t < t1; t++) {
            const float4 sp = s4[t];
            D = fminf(D, sqd(px - sp.x, py - sp.y, pz - sp.z));
        }
    }
    smin[sl][p] = D;
    __syncthreads();
    if (sl == 0 && x < n) {
#pragma unroll
        for (int w = 1; w < VER_T / VER_P; w++) D = fminf(D, smin[w][p]);
        reinterpret_cast<float *>(pts + inv[start_n + x])[3] = D;
    }
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


__device__ __forceinline__ void lds_barrier() {
    // LDS traffic of this wave retired, then the workgroup barrier; global stores stay in flight
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// branch-free variant (two DPP passes), so that several independent reductions can be interleaved
__device__ __forceinline__ KeyMax wave_key_max_bf(unsigned long long v) {
    const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
    const unsigned mh = wave_max_u32(hi);
    const unsigned ml = wave_max_u32(hi == mh ? lo : 0u);
    KeyMax r;
    r.key = ((unsigned long long)mh << 32) | ml;
    r.lane = __ffsll((unsigned long long)__ballot(hi == mh && lo == ml)) - 1;
    return r;
}

// STAMP: diagnostic build only (P2_FPS_STAMPS=1): per-wave cycle sums of the step phases -> dbg; P2_FPS_TRACE=file
// also dumps the absolute phase times of every wave for FPS_TRACE_STEPS steps (tools/fps_trace.py)
constexpr int FPS_TRACE_STEPS = 512;
template <int NBL, int NW, bool STAMP = false, bool B64 = false>
__global__ __launch_bounds__(NW * 64) void fps_bucket_kernel(int Bref, int log2B, int BSZ_arg, const float *__restrict__ xyz,
                                                             const int *__restrict__ offset, const int *__restrict__ new_offset,
                                                             float4 *__restrict__ pts, const unsigned *__restrict__ rank,
                                                             const int *__restrict__ prev_idx, const int *__restrict__ prev_offset,
                                                             const int *__restrict__ verified, int *__restrict__ idx,
                                                             unsigned long long *__restrict__ dbg = nullptr) {
    constexpr int NT = NW * 64;
    unsigned long long c_test = 0, c_red = 0, c_bar = 0, c_fin = 0, n_upd = 0, t_a = 0, t_b = 0;
    unsigned long long rt0 = 0, ct0 = 0;
    if (STAMP) { rt0 = __builtin_amdgcn_s_memrealtime(); ct0 = __builtin_amdgcn_s_memtime(); }
    __shared__ unsigned long long wkey[2][NW];
    __shared__ float4 wbest[2][NW];

    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int start_n = bid == 0 ? 0 : offset[bid - 1], end_n = offset[bid];
    const int start_m = bid == 0 ? 0 : new_offset[bid - 1], end_m = new_offset[bid];
    if (end_n <= start_n) {
        for (int j = start_m + tid; j < end_m; j += NT) idx[j] = start_n;
        return;
    }
    const int n = end_n - start_n;
    const int BSZ = B64 ? 64 : BSZ_arg;  // 64-point buckets (every cloud up to 131072 points) are a compile-time fact:
                                         // the generic paths below then do not exist in the code
    const int nb = (n + BSZ - 1) / BSZ;

    // samples inherited from the previous call on this state
    int done = 0;
    if (prev_idx) {
        const int ps = bid == 0 ? 0 : prev_offset[bid - 1], pe = prev_offset[bid];
        done = min(pe - ps, end_m - start_m);
        for (int t = tid; t < done; t += NT) idx[start_m + t] = prev_idx[ps + t];
    }
    // samples verified to be the identity prefix 0,1,2,... (fps_verify_*); the min-dist field was rebuilt for them
    if (verified) {
        const int v = min(verified[bid], end_m - start_m);
        if (v > done) {
            for (int t = tid; t < v; t += NT) idx[start_m + t] = start_n + t;
            done = v;
        }
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
            const int bk = (s * 64 + l) * NW + wave;
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
            const float wx = rl(cx, km.lane), wy = rl(cy, km.lane), wz = rl(cz, km.lane);
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

    // the wave's best bucket, kept wave-uniform between steps: key, coordinates and which owned bucket holds
    // it (code = slot*64 + lane).  Keys only decrease, so it changes only when the holder itself is updated.
    unsigned long long wk = 0ull;
    float wx = 0.f, wy = 0.f, wz = 0.f;
    int whold = -1;
    bool recompute = true;

    // The owner lane (code & 63) of slot code >> 6 takes the bucket's new key and best point.  With 64-point buckets
    // this is five v_writelane_b32 (lane select in m0) under a wave-uniform slot test; the generic form masks lanes.
    // (On the step's critical path an instruction costs ~9 cycles, tools/fps_trace.py: count them.)
    auto update_regs = [&](int code, const KeyMax &km, float cx, float cy, float cz) {
        if (BSZ == 64) {
            const int ol = __builtin_amdgcn_readfirstlane(code & 63);
#pragma unroll
            for (int s = 0; s < NBL; s++)
                if ((code >> 6) == s) {
                    int lo = (int)(unsigned)key[s], hi = (int)(unsigned)(key[s] >> 32);
                    int ix = __float_as_int(bx[s]), iy = __float_as_int(by[s]), iz = __float_as_int(bz[s]);
                    // v_writelane_b32 with an SGPR source takes its lane select from m0 on gfx9 (one SGPR per VALU instruction on
                    // the constant bus; this clang has no __builtin_amdgcn_writelane).  m0 is SAVED and RESTORED inside the
                    // statement instead of being declared clobbered: no reserved register changes across it.
                    int m0_saved;
                    asm volatile("s_mov_b32 %5, m0\n\ts_mov_b32 m0, %11\n\t"
                                 "v_writelane_b32 %0, %6, m0\n\tv_writelane_b32 %1, %7, m0\n\tv_writelane_b32 %2, %8, m0\n\t"
                                 "v_writelane_b32 %3, %9, m0\n\tv_writelane_b32 %4, %10, m0\n\t"
                                 "s_mov_b32 m0, %5"
                                 : "+v"(lo), "+v"(hi), "+v"(ix), "+v"(iy), "+v"(iz), "=&s"(m0_saved)
                                 : "s"((int)(unsigned)km.key), "s"((int)(unsigned)(km.key >> 32)),
                                   "s"(__builtin_amdgcn_readfirstlane(__float_as_int(cx))), "s"(__builtin_amdgcn_readfirstlane(__float_as_int(cy))),
                                   "s"(__builtin_amdgcn_readfirstlane(__float_as_int(cz))), "s"(ol));
                    key[s] = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
                    bx[s] = __int_as_float(ix); by[s] = __int_as_float(iy); bz[s] = __int_as_float(iz);
                }
        } else {
#pragma unroll
            for (int s = 0; s < NBL; s++)
                if ((code >> 6) == s && lane == (code & 63)) { key[s] = km.key; bx[s] = cx; by[s] = cy; bz[s] = cz; }
        }
        if (code == whold) recompute = true;
    };

    for (int j = start_m + max(done, 1); j < end_m; j++) {
        const bool trace = STAMP && dbg && (j - start_m) >= 5000 && (j - start_m) < 5000 + FPS_TRACE_STEPS;
        unsigned long long *tr = nullptr;
        if (trace) tr = dbg + 16 * 8 + ((size_t)(j - start_m - 5000) * NW + wave) * 8;
        if (STAMP) t_a = __builtin_amdgcn_s_memtime();
        if (trace && lane == 0) tr[0] = t_a;
        unsigned long long hm[NBL];
        unsigned long long any = 0ull;
        int ntouched = 0;
#pragma unroll
        for (int s = 0; s < NBL; s++) {
            const float dx = fmaxf(fmaxf(mnx[s] - x1, x1 - mxx[s]), 0.f);
            const float dy = fmaxf(fmaxf(mny[s] - y1, y1 - mxy[s]), 0.f);
            const float dz = fmaxf(fmaxf(mnz[s] - z1, z1 - mxz[s]), 0.f);
            hm[s] = __ballot(sqd(dx, dy, dz) < __uint_as_float((unsigned)(key[s] >> 32)));
            any |= hm[s];
            ntouched += __popcll(hm[s]);
        }
        if (STAMP) n_upd += ntouched;
        if (trace && lane == 0) { tr[1] = __builtin_amdgcn_s_memtime(); tr[7] = ntouched; }
        if (ntouched == 1 && BSZ == 64) {
            // the common case: exactly one owned bucket to update, straight-line
            int code = -1;
#pragma unroll
            for (int s = 0; s < NBL; s++)
                if (hm[s]) code = s * 64 + __ffsll(hm[s]) - 1;
            // straight-line: lanes past the end of the cloud work on a copy of its last point (they can only tie with
            // that point itself), and the distance is stored whether it changed or not
            code = __builtin_amdgcn_readfirstlane(code);
            const int pos = min(start_n + (code * NW + wave) * 64 + lane, end_n - 1);
            const float4 p = pts[pos];
            const unsigned rk = rank[pos];
            const float d2 = fminf(sqd(p.x - x1, p.y - y1, p.z - z1), p.w);
            reinterpret_cast<float *>(pts + pos)[3] = d2;
            const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | rk;
            const KeyMax km = wave_key_max(k);
            update_regs(code, km, rl(p.x, km.lane), rl(p.y, km.lane), rl(p.z, km.lane));
            any = 0ull;
        }
        if (ntouched == 2 && B64) {
            // two owned buckets (the busiest wave of most steps): both loads first, then the two reductions, no loop
            int c2[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                int c = -1;
                bool taken = false;
#pragma unroll
                for (int s = 0; s < NBL; s++) {
                    const bool take = !taken && hm[s] != 0ull;
                    c = take ? s * 64 + (int)__ffsll(hm[s]) - 1 : c;
                    hm[s] = take ? (hm[s] & (hm[s] - 1)) : hm[s];
                    taken = taken || take;
                }
                c2[u] = __builtin_amdgcn_readfirstlane(c);
            }
            const int pa = min(start_n + (c2[0] * NW + wave) * 64 + lane, end_n - 1);
            const int pb = min(start_n + (c2[1] * NW + wave) * 64 + lane, end_n - 1);
            const float4 qa = pts[pa], qb = pts[pb];
            const unsigned ra = rank[pa], rb = rank[pb];
            const float da = fminf(sqd(qa.x - x1, qa.y - y1, qa.z - z1), qa.w);
            const float db = fminf(sqd(qb.x - x1, qb.y - y1, qb.z - z1), qb.w);
            reinterpret_cast<float *>(pts + pa)[3] = da;
            reinterpret_cast<float *>(pts + pb)[3] = db;
            const KeyMax ka = wave_key_max(((unsigned long long)__float_as_uint(da) << 32) | ra);
            update_regs(c2[0], ka, rl(qa.x, ka.lane), rl(qa.y, ka.lane), rl(qa.z, ka.lane));
            const KeyMax kb = wave_key_max(((unsigned long long)__float_as_uint(db) << 32) | rb);
            update_regs(c2[1], kb, rl(qb.x, kb.lane), rl(qb.y, kb.lane), rl(qb.z, kb.lane));
            any = 0ull;
        }
        if (ntouched == 3 && B64) {
            // three owned buckets: the same straight-line form
            int c3[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                int c = -1;
                bool taken = false;
#pragma unroll
                for (int s = 0; s < NBL; s++) {
                    const bool take = !taken && hm[s] != 0ull;
                    c = take ? s * 64 + (int)__ffsll(hm[s]) - 1 : c;
                    hm[s] = take ? (hm[s] & (hm[s] - 1)) : hm[s];
                    taken = taken || take;
                }
                c3[u] = __builtin_amdgcn_readfirstlane(c);
            }
            int pp[3];
            float4 qq[3];
            unsigned rr[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                pp[u] = min(start_n + (c3[u] * NW + wave) * 64 + lane, end_n - 1);
                qq[u] = pts[pp[u]];
                rr[u] = rank[pp[u]];
            }
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const float d2 = fminf(sqd(qq[u].x - x1, qq[u].y - y1, qq[u].z - z1), qq[u].w);
                reinterpret_cast<float *>(pts + pp[u])[3] = d2;
                const KeyMax km = wave_key_max(((unsigned long long)__float_as_uint(d2) << 32) | rr[u]);
                update_regs(c3[u], km, rl(qq[u].x, km.lane), rl(qq[u].y, km.lane), rl(qq[u].z, km.lane));
            }
            any = 0ull;
        }
        while (any) {
            // next (up to) four touched buckets across all slots: code = slot*64 + owner lane, -1 = none (scalar
            // selects, no branches)
            int code[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int c = -1;
                bool taken = false;
#pragma unroll
                for (int s = 0; s < NBL; s++) {
                    const bool take = !taken && hm[s] != 0ull;
                    c = take ? s * 64 + (int)__ffsll(hm[s]) - 1 : c;
                    hm[s] = take ? (hm[s] & (hm[s] - 1)) : hm[s];
                    taken = taken || take;
                }
                code[u] = c;
            }
            any = 0ull;
#pragma unroll
            for (int s = 0; s < NBL; s++) any |= hm[s];
            if (BSZ == 64) {
                // all loads first (kept in flight together), then one reduction per touched bucket; straight-line
                // per bucket like the single-bucket path (clamped lanes, distance stored unconditionally)
                float4 p[4];
                unsigned rk[4];
                int pos[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    code[u] = __builtin_amdgcn_readfirstlane(code[u]);
                    pos[u] = min(start_n + (max(code[u], 0) * NW + wave) * 64 + lane, end_n - 1);
                    if (code[u] >= 0) { p[u] = pts[pos[u]]; rk[u] = rank[pos[u]]; }  // wave-uniform
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (code[u] < 0) continue;  // wave-uniform: absent entries cost nothing
                    const float d2 = fminf(sqd(p[u].x - x1, p[u].y - y1, p[u].z - z1), p[u].w);
                    reinterpret_cast<float *>(pts + pos[u])[3] = d2;
                    const KeyMax km = wave_key_max(((unsigned long long)__float_as_uint(d2) << 32) | rk[u]);
                    update_regs(code[u], km, rl(p[u].x, km.lane), rl(p[u].y, km.lane), rl(p[u].z, km.lane));
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (code[u] < 0) continue;  // wave-uniform
                    const int bk = code[u] * NW + wave;
                    const int p0 = start_n + bk * BSZ, p1 = min(p0 + BSZ, end_n);
                    unsigned long long best = 0ull;
                    float cx = 0.f, cy = 0.f, cz = 0.f;
                    for (int pos = p0 + lane; pos < p1; pos += 64) {
                        const float4 q = pts[pos];
                        const float d = sqd(q.x - x1, q.y - y1, q.z - z1);
                        const float d2 = fminf(d, q.w);
                        if (d2 != q.w) reinterpret_cast<float *>(pts + pos)[3] = d2;
                        const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | rank[pos];
                        if (k > best) { best = k; cx = q.x; cy = q.y; cz = q.z; }
                    }
                    const KeyMax km = wave_key_max(best);
                    update_regs(code[u], km, rl(cx, km.lane), rl(cy, km.lane), rl(cz, km.lane));
                }
            }
        }
        if (STAMP) { t_b = __builtin_amdgcn_s_memtime(); c_test += t_b - t_a; t_a = t_b; }
        if (trace && lane == 0) tr[2] = t_b;
        if (recompute) {  // wave-uniform
            unsigned long long mk = key[0];
            float mx_ = bx[0], my_ = by[0], mz_ = bz[0];
            int ms = 0;
#pragma unroll
            for (int s = 1; s < NBL; s++)
                if (key[s] > mk) { mk = key[s]; mx_ = bx[s]; my_ = by[s]; mz_ = bz[s]; ms = s; }
            const KeyMax wm = wave_key_max(mk);
            wk = wm.key;
            wx = rl(mx_, wm.lane); wy = rl(my_, wm.lane); wz = rl(mz_, wm.lane);
            whold = __builtin_amdgcn_readlane(ms, wm.lane) * 64 + wm.lane;
            recompute = false;
            if (lane == 0) {
                wkey[0][wave] = wk;
                wbest[0][wave] = make_float4(wx, wy, wz, 0.f);
            }
        }
        if (STAMP) { t_b = __builtin_amdgcn_s_memtime(); c_red += t_b - t_a; t_a = t_b; }
        if (trace && lane == 0) tr[3] = t_b;
        lds_barrier();  // every wave's slot is current
        unsigned long long win_key = 0ull;
        if (STAMP) { t_b = __builtin_amdgcn_s_memtime(); c_bar += t_b - t_a; t_a = t_b; }
        if (trace && lane == 0) tr[4] = t_b;
        if (wave == 0) {
            // arg-max over the NW slots (one wave, so the others do not compete for issue slots)
            const int src = lane < NW ? lane : 0;
            const unsigned long long gk = lane < NW ? wkey[0][src] : 0ull;
            const float4 gc = wbest[0][src];
            KeyMax gm;
            if (NW <= 16) {
                const unsigned hi = (unsigned)(gk >> 32), lo = (unsigned)gk;
                const unsigned mh = row0_max_u32(hi);
                const unsigned long long tied = __ballot(hi == mh && lane < NW);
                if (__popcll(tied) == 1) {
                    gm.lane = __ffsll(tied) - 1;
                    gm.key = ((unsigned long long)mh << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, gm.lane);
                } else {
                    const unsigned ml = row0_max_u32((hi == mh && lane < NW) ? lo : 0u);
                    gm.key = ((unsigned long long)mh << 32) | ml;
                    gm.lane = __ffsll((unsigned long long)__ballot(hi == mh && lo == ml && lane < NW)) - 1;
                }
            } else {
                gm = wave_key_max(gk);
            }
            if (lane == 0) wbest[1][0] = make_float4(rl(gc.x, gm.lane), rl(gc.y, gm.lane), rl(gc.z, gm.lane), 0.f);
            win_key = gm.key;
        }
        if (trace && lane == 0) tr[5] = __builtin_amdgcn_s_memtime();
        lds_barrier();  // the new sample is published
        {
            const float4 ns = wbest[1][0];
            x1 = ns.x; y1 = ns.y; z1 = ns.z;
        }
        // the sample's index (key -> position, a dozen instructions and a store) is nobody's input: it is written
        // behind the barrier, off the path the 15 other waves wait on
        if (wave == 0 && lane == 0) idx[j] = start_n + rel_of(win_key, Bref, log2B);
        if (STAMP) { t_b = __builtin_amdgcn_s_memtime(); c_fin += t_b - t_a; }
        if (trace && lane == 0) tr[6] = t_b;
    }
    if (STAMP && dbg && lane == 0) {
        unsigned long long *o = dbg + (blockIdx.x * NW + wave) * 8;
        o[0] = c_test; o[1] = 0; o[2] = c_red; o[3] = c_bar; o[4] = c_fin; o[5] = n_upd;
        o[6] = __builtin_amdgcn_s_memtime() - ct0; o[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    }
}

static size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static int bits_for(int b) {
    int r = 1;
    while ((1 << r) < b) r++;
    return r;
}
constexpr int FPS_HEAD_MAX = 4096;  // samples of a cloud the round sampler may leave to the step-by-step kernel (P2_FPS_HEAD)

static size_t fps_cub_bytes(int b, int N) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (const int *)nullptr, (int *)nullptr, N, 0, 32 + bits_for(b), (hipStream_t) nullptr);
    return bytes;
}

// request of the sampling chain's head (fps_bucket_launch): the first `head` samples of every cloud
__global__ void fps_head_offsets_kernel(int b, int head, const int *__restrict__ new_offset, int *__restrict__ head_offset) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int acc = 0;
    for (int i = 0; i < b; i++) {
        const int m = new_offset[i] - (i ? new_offset[i - 1] : 0);
        acc += min(m, head);
        head_offset[i] = acc;
    }
}

struct FpsResume {
    const int *prev_idx = nullptr;
    const int *prev_offset = nullptr;
    bool unordered = false;  // pointops2_set_fps_hint: the caller knows the cloud is in no selection order (no identity-prefix probe)
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
    int *inv = (int *)p; p += f4;                                   // original index -> sampler order (persistent)
    unsigned long long *thr = (unsigned long long *)p; p += f8;    // verification thresholds
    int *first_bad = (int *)p; p += al((size_t)b * 4);
    void *xchg = p; p += al((size_t)b * LZ_XCHG);                  // where the round sampler's workgroups meet (fps_lazy.hip)
    int *head_offset = (int *)p; p += al((size_t)b * 4);            // the chain's head, sampled step by step (below)
    int *head_idx = (int *)p; p += al((size_t)b * FPS_HEAD_MAX * 4);
    void *cub_tmp = p;
    size_t cub_bytes = w.bytes - (size_t)(p - reinterpret_cast<char *>(w.ptr));
    FpsResume rs = fps_resume();
    fps_resume() = FpsResume();
    if (rs.prev_idx == nullptr) {
        hipLaunchKernelGGL(fps_bbox_kernel, dim3(b), dim3(BBOX_T), 0, st, xyz, offset, bbox);
        hipLaunchKernelGGL(fps_morton_kernel, dim3(div_up(N_total, 256)), dim3(256), 0, st, N_total, b, xyz, offset, bbox, keys_in, vals_in);
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(cub_tmp, cub_bytes, keys_in, keys_out, (const int *)vals_in, sorig, N_total, 0,
                                                          32 + bits_for(b), st);
        if (e != hipSuccess) { set_error(hipGetErrorString(e)); return true; }
        hipLaunchKernelGGL(fps_gather_kernel, dim3(div_up(N_total, 256)), dim3(256), 0, st, N_total, b, Bref, log2B, xyz, offset, sorig, pts, rank, inv);
    }
    // identity-prefix verification (exact; see above): a cheap probe of the first 64 steps, then everything
    static const bool no_verify = getenv("P2_FPS_NO_VERIFY") != nullptr;
    const int *verified = nullptr;
    if (!no_verify && !rs.unordered) {
        hipLaunchKernelGGL(fps_verify_init_kernel, dim3(div_up(b, 64)), dim3(64), 0, st, b, offset, new_offset, first_bad);
        hipLaunchKernelGGL(fps_verify_threshold_kernel, dim3(1, b), dim3(VER_T), 0, st, Bref, log2B, 1, xyz, offset, new_offset, first_bad, thr);
        hipLaunchKernelGGL(fps_verify_scan_kernel, dim3(div_up(n, VER_P), b), dim3(VER_T), 0, st, Bref, log2B, 1, 64, xyz, offset, new_offset, thr, first_bad);
        hipLaunchKernelGGL(fps_verify_threshold_kernel, dim3(div_up(n, VER_P), b), dim3(VER_T), 0, st, Bref, log2B, 64, xyz, offset, new_offset, first_bad, thr);
        hipLaunchKernelGGL(fps_verify_scan_kernel, dim3(div_up(n, VER_P), b), dim3(VER_T), 0, st, Bref, log2B, 64, 0x7fffffff, xyz, offset, new_offset, thr, first_bad);
        hipLaunchKernelGGL(fps_rebuild_kernel, dim3(div_up(n, VER_P), b), dim3(VER_T), 0, st, xyz, offset, rs.prev_offset, first_bad, inv, pts);
        verified = first_bad;
    }
    const int BSZ = 64 * div_up(n, 64 * FPS_MAX_BUCKETS);
    const int nbuckets = div_up(n, BSZ);
    static const int nw_env = getenv("P2_FPS_WAVES") ? atoi(getenv("P2_FPS_WAVES")) : 0;
    const int NWsel = nw_env == 8 ? 8 : 16;
    const int per_lane = div_up(nbuckets, NWsel * 64);
#define P2_FPS_LAUNCH(NBL_, NW_, STAMP_, DBG_, NEWOFF_, PIDX_, POFF_, IDX_)                                                 \
    do {                                                                                                                    \
        if (BSZ == 64)                                                                                                      \
            hipLaunchKernelGGL((fps_bucket_kernel<NBL_, NW_, STAMP_, true>), dim3(b), dim3(NW_ * 64), 0, st, Bref, log2B, BSZ, xyz, \
                               offset, NEWOFF_, pts, rank, PIDX_, POFF_, verified, IDX_, DBG_);                             \
        else                                                                                                                \
            hipLaunchKernelGGL((fps_bucket_kernel<NBL_, NW_, STAMP_, false>), dim3(b), dim3(NW_ * 64), 0, st, Bref, log2B, BSZ, xyz, \
                               offset, NEWOFF_, pts, rank, PIDX_, POFF_, verified, IDX_, DBG_);                             \
    } while (0)
#define P2_FPS_STEPWISE(NEWOFF_, PIDX_, POFF_, IDX_)                                                                       \
    do {                                                                                                                    \
        if (NWsel == 8) {                                                                                                   \
            if (per_lane <= 1) P2_FPS_LAUNCH(1, 8, false, nullptr, NEWOFF_, PIDX_, POFF_, IDX_);                            \
            else if (per_lane <= 2) P2_FPS_LAUNCH(2, 8, false, nullptr, NEWOFF_, PIDX_, POFF_, IDX_);                       \
            else P2_FPS_LAUNCH(4, 8, false, nullptr, NEWOFF_, PIDX_, POFF_, IDX_);                                          \
        } else {                                                                                                            \
            if (per_lane <= 1) P2_FPS_LAUNCH(1, 16, false, nullptr, NEWOFF_, PIDX_, POFF_, IDX_);                           \
            else P2_FPS_LAUNCH(2, 16, false, nullptr, NEWOFF_, PIDX_, POFF_, IDX_);                                         \
        }                                                                                                                   \
    } while (0)
    // Round-based sampler (fps_lazy.hip) on the same state; P2_FPS_STEPWISE=1 keeps the step-by-step kernel.  The first few
    // hundred samples of a fresh chain interact so strongly that a round decides one or two of them (and a round costs what
    // ~30 dependent steps cost): the HEAD of the chain is therefore sampled step by step and the rounds resume from it -
    // same state conventions, same sequence.
    static const bool stepwise = getenv("P2_FPS_STEPWISE") != nullptr;
    if (!stepwise && fps_lazy_groups(n) > 0) {
        static const int head_env = getenv("P2_FPS_HEAD") ? atoi(getenv("P2_FPS_HEAD")) : 256;
        const int head = std::min(std::max(head_env, 0), FPS_HEAD_MAX);
        const int *pidx = rs.prev_idx, *poff = rs.prev_offset;
        if (pidx == nullptr && head > 0 && n >= 8192) {
            hipLaunchKernelGGL(fps_head_offsets_kernel, dim3(1), dim3(64), 0, st, b, head, new_offset, head_offset);
            P2_FPS_STEPWISE(head_offset, (const int *)nullptr, (const int *)nullptr, head_idx);
            held_cus_note(st, b);  // (one workgroup per cloud holds a CU for the head's ~1 ms)
            pidx = head_idx;
            poff = head_offset;
        }
        fps_lazy_launch(b, n, Bref, log2B, xyz, offset, new_offset, pts, rank, pidx, poff, verified, idx, xchg, st);
        return true;
    }
    if (getenv("P2_FPS_STAMPS") && nbuckets > 1024) {  // diagnostic only: synchronous, prints phase shares to stderr
        unsigned long long *dbg = nullptr, host[16 * 8];
        const size_t trace_words = (size_t)FPS_TRACE_STEPS * 16 * 8;
        (void)hipMalloc(&dbg, (sizeof(host) + trace_words * 8) * b);
        (void)hipMemset(dbg, 0, (sizeof(host) + trace_words * 8) * b);
        if (NWsel == 8) P2_FPS_LAUNCH(4, 8, true, dbg, new_offset, rs.prev_idx, rs.prev_offset, idx); else P2_FPS_LAUNCH(2, 16, true, dbg, new_offset, rs.prev_idx, rs.prev_offset, idx);
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(host, dbg, sizeof(host), hipMemcpyDeviceToHost);
        if (const char *tf = getenv("P2_FPS_TRACE")) {
            unsigned long long *tbuf = (unsigned long long *)malloc(trace_words * 8);
            (void)hipMemcpy(tbuf, dbg + 16 * 8, trace_words * 8, hipMemcpyDeviceToHost);
            if (FILE *f = fopen(tf, "wb")) { fwrite(tbuf, 8, trace_words, f); fclose(f); }
            free(tbuf);
        }
        (void)hipFree(dbg);
        for (int w = 0; w < NWsel; w++)
            fprintf(stderr, "[fps stamps] wave %2d: test+update %llu reduce %llu barrier %llu final %llu | updates %llu | cycles %llu realtime(100MHz) %llu -> %.0f MHz\n",
                    w, host[w * 8 + 0], host[w * 8 + 2], host[w * 8 + 3], host[w * 8 + 4], host[w * 8 + 5], host[w * 8 + 6], host[w * 8 + 7],
                    host[w * 8 + 7] ? 100.0 * host[w * 8 + 6] / host[w * 8 + 7] : 0.0);
        return true;
    }
    P2_FPS_STEPWISE(new_offset, rs.prev_idx, rs.prev_offset, idx);
    held_cus_note(st, b);
#undef P2_FPS_STEPWISE
#undef P2_FPS_LAUNCH
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
    return al((size_t)N * 16) + 4 * al((size_t)N * 4) + 3 * al((size_t)N * 8) + al((size_t)b * 6 * 4) + al((size_t)b * 4) + al((size_t)b * LZ_XCHG) + al((size_t)b * 4) + al((size_t)b * FPS_HEAD_MAX * 4) +
           al(fps_cub_bytes(b, N));
}

void pointops2_set_fps_resume(const int *prev_idx, const int *prev_offset) {
    fps_resume().prev_idx = prev_idx;
    fps_resume().prev_offset = prev_offset;
}

void pointops2_set_fps_hint(int unordered) { fps_resume().unordered = unordered != 0; }

}  // extern "C"
