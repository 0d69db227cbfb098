// Shared pieces of the window-centric ("cell") attention kernels (cell_attn.hip: VALU walkers; cell_attn_mfma.hip: matrix-core
// forward): cross-lane helpers, buffer addressing, the task dealing of a cell plan and the persistent grid.
#pragma once
#include "rpe_common.h"
#include <algorithm>
#include <cstdlib>

namespace p2 {

typedef float __attribute__((ext_vector_type(4))) f32x4c;

#ifndef CA_NP_OVERRIDE
#define CA_NP_OVERRIDE 8
#endif
constexpr int CA_NP = CA_NP_OVERRIDE;  // forward: passes of 16 keys a lane keeps in registers (128 keys per chunk)
constexpr int CA_NP_BWD = 3;  // backward: 48 keys per chunk (key rows AND their gradient accumulators in registers; chunks simply add up)
#ifndef CA_WAVES_OVERRIDE
#define CA_WAVES_OVERRIDE 12
#endif
constexpr int CA_WAVES = CA_WAVES_OVERRIDE;  // waves per workgroup (one head's three tables in LDS per workgroup)
#ifndef CA_WAVES_BWD_OVERRIDE
#define CA_WAVES_BWD_OVERRIDE 12
#endif
constexpr int CA_WAVES_BWD = CA_WAVES_BWD_OVERRIDE;  // backward workgroup

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
// N consecutive dwords per lane as ONE (N <= 4) or two vector-memory instructions: the walkers below are bound by the
// number of memory instructions a CU can issue, not by bytes
typedef unsigned u32x2c __attribute__((ext_vector_type(2)));
typedef unsigned u32x3c __attribute__((ext_vector_type(3)));
typedef unsigned u32x4c __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void bload_words(rsrc_t r, int off, unsigned (&w)[N]) {
    static_assert(N == 2 || N == 3 || N == 4 || N == 6 || N == 8, "pass counts of dispatch_passes");
    if constexpr (N == 2) {
        const u32x2c a = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        w[0] = a.x; w[1] = a.y;
    } else if constexpr (N == 3) {
        const u32x3c a = __builtin_amdgcn_raw_buffer_load_b96(r, off, 0, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z;
    } else {
        const u32x4c a = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w;
        if constexpr (N == 6) {
            const u32x2c b = __builtin_amdgcn_raw_buffer_load_b64(r, off + 16, 0, 0);
            w[4] = b.x; w[5] = b.y;
        } else if constexpr (N == 8) {
            const u32x4c b = __builtin_amdgcn_raw_buffer_load_b128(r, off + 16, 0, 0);
            w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
        }
    }
}
template <int N>
__device__ __forceinline__ void bload_floats(rsrc_t r, int off, float (&f)[N]) {
    unsigned w[N];
    bload_words<N>(r, off, w);
#pragma unroll
    for (int t = 0; t < N; t++) f[t] = __uint_as_float(w[t]);
}
// stores the first `nvalid` of a lane's N floats (a lane whose slots are all inside the row: one or two wide stores;
// the one lane that straddles the row's end: dword stores, so that the next row's entries stay intact)
template <int N>
__device__ __forceinline__ void bstore_floats(rsrc_t r, int off, const float (&f)[N], int nvalid) {
    if (nvalid >= N) {
        if constexpr (N == 2) {
            __builtin_amdgcn_raw_buffer_store_b64(u32x2c{__float_as_uint(f[0]), __float_as_uint(f[1])}, r, off, 0, 0);
        } else if constexpr (N == 3) {
            __builtin_amdgcn_raw_buffer_store_b96(u32x3c{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2])}, r, off, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b128(u32x4c{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])}, r,
                                                   off, 0, 0);
            if constexpr (N == 6)
                __builtin_amdgcn_raw_buffer_store_b64(u32x2c{__float_as_uint(f[4]), __float_as_uint(f[5])}, r, off + 16, 0, 0);
            else if constexpr (N == 8)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4c{__float_as_uint(f[4]), __float_as_uint(f[5]), __float_as_uint(f[6]), __float_as_uint(f[7])},
                                                       r, off + 16, 0, 0);
        }
    } else {
#pragma unroll
        for (int t = 0; t < N; t++)
            if (t < nvalid) bstore_f32(r, off + 4 * t, f[t]);
    }
}


struct CellTask {
    int qs, nq, kb, nk, pbase;
};
// (everything about a cell is wave-uniform: held in scalar registers)
__device__ __forceinline__ CellTask cell_task(const pointops2_cell_plan &pl, int task) {
    // (an explicit task list holds cell ids; otherwise `task` is a position of the size-sorted cell list)
    const int cell = __builtin_amdgcn_readfirstlane(pl.task_list != nullptr ? pl.task_list[task] : pl.cell_perm[task]);
    CellTask t;
    t.qs = __builtin_amdgcn_readfirstlane(pl.cell_qstart[cell]);
    t.nq = __builtin_amdgcn_readfirstlane(pl.cell_qstart[cell + 1]) - t.qs;
    t.kb = __builtin_amdgcn_readfirstlane(pl.cell_kbase[cell]);
    t.nk = __builtin_amdgcn_readfirstlane(pl.cell_kbase[cell + 1]) - t.kb;
    t.pbase = __builtin_amdgcn_readfirstlane(pl.cell_pbase[cell]);
    return t;
}
// Tasks are sorted by decreasing tile size and dealt to the resident waves in boustrophedon order (round r forwards,
// round r+1 backwards), so that no wave collects the largest task of every round.
__device__ __forceinline__ int snake_task(int round, int slot, int slots) { return round * slots + ((round & 1) ? slots - 1 - slot : slot); }
// A launch may work on a share of the cells only (pointops2_cell_plan.task_first / task_step: one scene over several ranks):
// the i-th task of the launch is cell_perm[first + i * step]
__device__ __forceinline__ int share_count(const pointops2_cell_plan &pl, int n) {
    if (pl.task_list != nullptr) return __builtin_amdgcn_readfirstlane(pl.task_count[0]);
    const int step = pl.task_step > 1 ? pl.task_step : 1, first = pl.task_step > 1 ? pl.task_first : 0;
    return n > first ? (n - first + step - 1) / step : 0;
}
__device__ __forceinline__ int share_task(const pointops2_cell_plan &pl, int i) {
    if (pl.task_list != nullptr) return i;
    return pl.task_step > 1 ? pl.task_first + i * pl.task_step : i;
}


// Persistent grid of the cell walkers: `per_cu` workgroups for every CU that is FREE, over all heads; never more waves than
// tasks.  Tasks are dealt by position, so a workgroup that has to wait for a CU serves its whole share late and the kernel
// takes twice as long (tools/cell_trace.py: 16 CUs held -> 31 of 255 workgroups start when the others finish).  The one
// long-running kernel of this library is the round sampler (16 workgroups per cloud for milliseconds): its launches are noted
// (common.h, held_cus_*), and while any of them has not finished the grid leaves its shader engines room: workgroups go to
// the 32 shader engines (8 CUs each) in turn, so one CU per engine is left out (usable_cus(), common.h).  Measured, stage-0
// forward: 265 us alone, 451 us beside 16 held CUs, 314 us with a grid of 7 per engine (283 us alone with that grid).
// (Other designs measured: a work queue - contended device-scope atomics, 1.2x slower alone; several workgroups per CU taking
// contiguous task ranges in dispatch order - the table staging per workgroup and idle waves cost 1.1-1.4x alone; fewer waves
// per workgroup with more workgroups per CU - 1.1-1.5x slower.)
static int cell_grid_x(int per_cu, int tasks, int h, int waves) {
    const int cap = max(1, per_cu * usable_cus() / max(h, 1));
    return max(1, min(cap, div_up(tasks, waves)));
}

// cell_attn_mfma.hip: the forward on the matrix cores (fp32 operands, L <= 80); false = does not apply
bool cell_fwd_mfma_launch(const pointops2_cell_plan *plan, int h, int L, const float *q, const float *k, const float *v, const float *table_q,
                          const float *table_k, const float *table_v, float *out, float *pbuf);

}  // namespace p2
