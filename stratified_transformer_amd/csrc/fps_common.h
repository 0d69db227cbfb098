// Helpers shared by the exact FPS kernels (fps_bucket.hip, fps_lazy.hip).
#pragma once
#include "common.h"

namespace p2 {

// the reference's squared distance (sampling_cuda_kernel.cu:52-53) with the fma chain nvcc contracts it to
__device__ __forceinline__ float sqd(float dx, float dy, float dz) {
    return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
}
// key = (min-dist bits << 32) | tie rank: its unsigned maximum is the reference's winner (launch-geometry tie rule, sampling.hip)
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
__device__ __forceinline__ float rl(float v, int lane) {  // value of a wave-uniform lane, no LDS round trip
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// fps_lazy.hip: the round-based sampler on the state fps_bucket_launch prepares; same arguments as fps_bucket_kernel plus the
// exchange area its workgroups meet in (b * LZ_XCHG bytes of the workspace) and the largest cloud of the batch
constexpr int LZ_CAP = 512;    // candidates / accepted samples per round (1024 / 704: fewer rounds, each dearer: 14.3 ms against 12.3)
constexpr int LZ_GMAX = 16;    // workgroups per batch element, at most
constexpr int LZ_NSLOT = 14;   // buckets of 64 points a wave keeps in registers (16 waves per workgroup)
int fps_lazy_groups(int n_max);  // workgroups the round sampler would use for a cloud of n_max points; 0: too large for it
constexpr size_t LZ_XCHG = 64 + 2 * LZ_GMAX * 32 + 2 * LZ_GMAX * LZ_CAP * 20;
void fps_lazy_launch(int b, int n_max, int Bref, int log2B, const float *xyz, const int *offset, const int *new_offset, float4 *pts, const unsigned *rank,
                     const int *prev_idx, const int *prev_offset, const int *verified, int *idx, void *xchg, hipStream_t st);

}  // namespace p2
