// Shared plumbing of libpointops2_hip.so (gfx950 only).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/pointops2_hip.h"

namespace p2 {

// ---- per-thread launch state (pointops2_set_stream / _set_table_rows / _set_csc) ----
struct LaunchState {
    hipStream_t stream = nullptr;
    const char *error = nullptr;
    int table_rows = 0;
    const int *csc_offsets = nullptr;
    const int *csc_pair = nullptr;
    const int *csc_query = nullptr;
    int total_points = 0;  // pointops2_set_point_count: N of the next furthestsampling / knnquery call (0 = unknown)
    int batch_count = 0;   // pointops2_set_batch_count: b of the next knnquery call (0 = unknown)
    int key_rows = 0;      // pointops2_set_key_rows: rows of k / v when they differ from the CSR's query rows (0 = same)
    const int *row_order = nullptr;  // pointops2_set_row_order: the rows in an order that keeps neighbours together (nullptr = by index)
    int row_order_n = 0;             // ... and the row count it was built for
};

// scratch memory lent by the caller (pointops2_set_workspace), thread-local
struct Workspace {
    void *ptr = nullptr;
    size_t bytes = 0;
};
Workspace &workspace();
LaunchState &state();

// fps_bucket.hip: returns false when the bucketed path does not apply (no workspace / unknown N)
bool fps_bucket_launch(int b, int n, int Bref, int log2B, const float *xyz, const int *offset, const int *new_offset,
                       int N_total, int *idx);

// fps_bucket.hip: per-batch-element bounding boxes [b][6] (min xyz, max xyz)
void launch_bbox(int b, const float *xyz, const int *offset, float *bbox, hipStream_t st);
// knn_grid.hip: returns false when the grid path does not apply (no workspace / unknown n, b / tiny problem)
bool knn_grid_launch(int m, int k, int n, int b, const float *xyz, const float *new_xyz, const int *offset,
                     const int *new_offset, int *idx, float *dist2);

inline void set_error(const char *msg) { state().error = msg; }

// A status word that kernels can set while they run (pinned host memory, mapped into every device): failures that only a running
// kernel can detect - the round sampler's grid barrier giving up, fps_lazy.hip - reach the host without a synchronisation of their
// own.  pointops2_last_error() reports a set word (and clears it) at the next library call after the kernel has run.
unsigned *async_status_word();               // device-visible address (nullptr if the allocation failed)
constexpr unsigned ASYNC_FPS_BARRIER_TIMEOUT = 1u;

inline bool check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(hipGetErrorString(e));
        return false;
    }
    return true;
}

// Fork / join over the library's own side streams: the independent kernels of ONE launcher call (e.g. the
// four kernels of the A2 backward write four different outputs) are enqueued side by side.  On the small
// stages each of them is latency-bound and fills a fraction of the chip, so they overlap almost fully.
// Everything is joined back into the caller's stream before the launcher returns: for the caller the call is
// still one in-order piece of work on its stream (stream capture sees an ordinary fork-join).
// Only worth it for small problems (`small`): kernels that fill the chip on their own gain nothing and lose a
// little to the extra events (stage 0 of the S3DIS config: A1 backward 377 -> 424 us; stages 2-3: -4 %).
// P2_NO_FORK=1 keeps every kernel on the caller's stream.
// The gain exists only with the runtime's default of 4 hardware queues (where side streams mostly share a queue and
// the fork is a cheap reordering): with GPU_MAX_HW_QUEUES=24 - which batches in flight need, DESIGN.md 5 - the forked
// kernels truly run side by side and every one of them gets slower (stage-2 block 775 -> 1007 us, stage 3 704 -> 920),
// so the launchers then keep to the caller's stream.  P2_FORK_MAX=<pair-heads> overrides the limit.
inline bool fork_worthwhile(int64_t pair_heads) {
    static const int64_t limit = [] {
        if (const char *e = getenv("P2_FORK_MAX")) return (int64_t)atoll(e);
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        return (q && atoi(q) > 4) ? (int64_t)0 : (int64_t)7000000;
    }();
    return pair_heads < limit;
}
class ForkJoin {
  public:
    ForkJoin(hipStream_t main, bool small);
    ~ForkJoin() { join(); }
    hipStream_t lane(int i);  // lane 0 = the caller's stream, lanes 1..3 = side streams (forked on first use)
    void join();
  private:
    hipStream_t main_;
    unsigned used_ = 0;
    bool enabled_;
};

// Long-running kernels of this library that hold CUs (the round sampler: 16 workgroups per cloud for milliseconds) note their
// launch here; the persistent-grid kernels ask how many such workgroups may still be running (an event per launch, polled).
void held_cus_note(hipStream_t st, int workgroups);
int held_cus_now();

constexpr int WAVE = 64;
constexpr int kNumCU = 256;  // MI355X (fallback when the device cannot be queried)
// compute units of the current device, queried once (a partitioned device has fewer); sizes the persistent grids
inline int num_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = kNumCU;
        return n;
    }();
    return cus;
}

// CUs a persistent grid should count on: workgroups are handed to the shader engines (8 CUs each) in turn, and one that has to
// wait for a CU held by a long-running kernel serves its whole static share late (cell_attn.hip, cell_grid_x).  While any noted
// launch has not finished, one CU per engine is left out (7 of 8: costs 7 % when nothing holds a CU after all, saves the 1.7-2x
// of a late tail when up to one CU per engine is held).  Not more than one: the question is asked when the kernel is ENQUEUED,
// and a host that runs a pass ahead of the device sees every sampler of the pass as pending - scaling the reserve with the
// pending count made a 20-step run 30 % slower than a 5-step one.
inline int usable_cus() {
    const int cus = num_cus(), engines = cus / 8 > 0 ? cus / 8 : 1;
    if (held_cus_now() <= 0) return cus;
    return engines * 7 < cus ? engines * 7 : cus;
}

// Dynamic LDS above the 64 KiB default needs an explicit opt-in per kernel (gfx950: 160 KiB per CU).
template <typename K>
inline void allow_big_lds(K kernel, size_t bytes) {
    // the attribute bounds dynamic + static LDS together: ask for what this launch needs, not the 160 KiB cap
    if (bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) set_error(hipGetErrorString(e));
    }
}

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// ---- rows in window order (pointops2_set_row_order) ----
// The pair walkers give a wave a row (a query, or a key of the transposed list) and gather the rows of its partners: 64-byte head
// rows of k / v / q / grad_out, and - by key - 12-byte snippets of the pair-indexed arrays.  In INDEX order neighbouring waves work
// on unrelated windows and every gather goes to memory (the operator path moves 12x its compulsory bytes).  In an order
// that keeps the rows of a window together (misc.hip, row_order_kernel: by the row's first partner, which is the lowest point id of
// its window) the waves that run at one time share their partners and the second-level cache serves the gathers.  Workgroups are
// handed to the eight XCDs in turn and every XCD has its own L2, so the order is cut into eight runs and workgroup b takes its slots
// from run b % 8.
inline const int *rows_in_order(int n_rows) {
    const LaunchState &s = state();
    return (s.row_order != nullptr && s.row_order_n == n_rows && n_rows >= 2048) ? s.row_order : nullptr;
}
// grid.x for a one-slot-per-wave launch (rows_per_wg waves): eight equal runs
inline int ordered_grid(int n_rows, int rows_per_wg) { return 8 * div_up(div_up(n_rows, 8), rows_per_wg); }
inline int64_t div_up64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers ----
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// the slots of one wave: by index (order == nullptr: slot = row, grid-strided) or along the eight runs of the row order
struct RowSlots {
    int pos, end, step;
    const int *order;
    __device__ __forceinline__ RowSlots(const int *order_, int n_rows, int waves_per_wg, int wave) : order(order_) {
        if (order_ != nullptr && (gridDim.x & 7) == 0) {
            const int per = (((n_rows + 7) >> 3) + waves_per_wg - 1) / waves_per_wg * waves_per_wg, run = blockIdx.x & 7;
            pos = run * per + (blockIdx.x >> 3) * waves_per_wg + wave;
            end = min((run + 1) * per, n_rows);
            step = (gridDim.x >> 3) * waves_per_wg;
        } else {
            order = nullptr;
            pos = blockIdx.x * waves_per_wg + wave;
            end = n_rows;
            step = gridDim.x * waves_per_wg;
        }
    }
    __device__ __forceinline__ bool more() const { return pos < end; }
    __device__ __forceinline__ int row() const { return order != nullptr ? order[pos] : pos; }
    __device__ __forceinline__ void next() { pos += step; }
};

__device__ __forceinline__ float4 ldg4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void stg4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

__device__ __forceinline__ float dot4(float4 a, float4 b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
    return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}
__device__ __forceinline__ float4 scale4(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }

// wave-wide unsigned max with DPP row operations (6 VALU + 1 readlane; a __shfl_xor butterfly costs 6
// dependent ds_bpermute round trips, and the 64-bit key would double that)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_step(unsigned v) {
    // old = 0 (identity of unsigned max) + bound_ctrl: rows masked off / lanes without a source read 0, which
    // lets the DPP combiner fold the move into v_max_u32_dpp (one instruction per step)
    const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, true);
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
// butterfly sum over the lanes whose ids differ only in bits [LO, HI) of the lane id
template <int LO_STRIDE, int HI_STRIDE>
__device__ __forceinline__ float xor_sum(float v) {
#pragma unroll
    for (int s = LO_STRIDE; s < HI_STRIDE; s <<= 1) v += __shfl_xor(v, s, 64);
    return v;
}
template <int LO_STRIDE, int HI_STRIDE>
__device__ __forceinline__ float4 xor_sum4(float4 v) {
#pragma unroll
    for (int s = LO_STRIDE; s < HI_STRIDE; s <<= 1) {
        v.x += __shfl_xor(v.x, s, 64);
        v.y += __shfl_xor(v.y, s, 64);
        v.z += __shfl_xor(v.z, s, 64);
        v.w += __shfl_xor(v.w, s, 64);
    }
    return v;
}

// Geometry of the per-query segment walkers: a head vector of D floats is spread over LPG lanes
// (one float4 each); a wave therefore holds PPW pairs of one query side by side.
template <int D>
struct Geo {
    static constexpr int LPG = D / 4;     // lanes per (pair, head)
    static constexpr int PPW = 64 / LPG;  // pairs per wave pass
};

}  // namespace p2
