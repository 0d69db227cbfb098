// I2: exact k-nearest-neighbour query, gfx950.
//
// Replaces lib/pointops2/src/knnquery/knnquery_cuda_kernel.cu:21-115 behind the same launcher.
//
// The reference's result is defined by its procedure, not only by "the k nearest": candidates are
// visited in index order, inserted into a k-max-heap under a strict '<' against the heap top and
// finally heap-sorted (unstable), so the order among equal distances depends on the heap history.
// This kernel replays exactly that procedure per query (one thread per query, same reheap /
// heap_sort), which makes idx and dist2 bit-identical, ties included.  What changes is the memory
// system around it:
//   - the heap lives in LDS as [slot][thread] columns (bank-conflict-free, no scratch spills; the
//     reference keeps float[100]+int[100] per thread in local memory);
//   - candidates are staged through LDS in coalesced 2048-point tiles shared by the workgroup and
//     read back as broadcast ds_read_b128 (the reference has every thread stream the cloud from
//     global memory).
#include "common.h"

namespace p2 {

constexpr int KNN_TILE = 2048;

template <int BS>
__global__ __launch_bounds__(BS) void knn_kernel(int m, int ns, const float *__restrict__ xyz,
                                                 const float *__restrict__ new_xyz, const int *__restrict__ offset,
                                                 const int *__restrict__ new_offset, int *__restrict__ idx,
                                                 float *__restrict__ dist2) {
    extern __shared__ float4 smem4[];
    float4 *tile = smem4;                                           // [KNN_TILE]
    float *hd = reinterpret_cast<float *>(smem4 + KNN_TILE);        // [ns][BS]
    int *hi = reinterpret_cast<int *>(hd + (size_t)ns * BS);        // [ns][BS]
    const int tid = threadIdx.x;
    const int pt = blockIdx.x * BS + tid;
    const bool active = pt < m;

    auto batch_of = [&](int q) {  // get_bt_idx, knnquery_cuda_kernel.cu:52-63 (terminates: q < m <= new_offset[b-1])
        int i = 0;
        while (!(q < new_offset[i])) i++;
        return i;
    };
    // candidate range of the whole workgroup: queries are batch-ordered
    const int q_first = blockIdx.x * BS, q_last = min(m, (blockIdx.x + 1) * BS) - 1;
    const int b_first = batch_of(q_first), b_last = batch_of(q_last);
    const int lo = b_first == 0 ? 0 : offset[b_first - 1];
    const int hi_end = offset[b_last];

    int start = 0, end = 0;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    if (active) {
        const int bt = batch_of(pt);
        start = bt == 0 ? 0 : offset[bt - 1];
        end = offset[bt];
        nx = new_xyz[pt * 3 + 0];
        ny = new_xyz[pt * 3 + 1];
        nz = new_xyz[pt * 3 + 2];
    }
    for (int i = 0; i < ns; i++) {
        hd[i * BS + tid] = 1e10f;
        hi[i * BS + tid] = start;
    }

    auto reheap = [&](int k) {  // :21-37
        int root = 0, child = 1;
        while (child < k) {
            if (child + 1 < k && hd[(child + 1) * BS + tid] > hd[child * BS + tid]) child++;
            const float dr = hd[root * BS + tid], dc = hd[child * BS + tid];
            if (dr > dc) return;
            hd[root * BS + tid] = dc;
            hd[child * BS + tid] = dr;
            const int ir = hi[root * BS + tid];
            hi[root * BS + tid] = hi[child * BS + tid];
            hi[child * BS + tid] = ir;
            root = child;
            child = root * 2 + 1;
        }
    };

    float top = 1e10f;  // register copy of the heap top
    for (int t0 = lo; t0 < hi_end; t0 += KNN_TILE) {
        const int cnt = min(KNN_TILE, hi_end - t0);
        __syncthreads();
        for (int t = tid; t < cnt; t += BS) {
            const float *p = xyz + (size_t)(t0 + t) * 3;
            tile[t] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        if (active) {
            const int a = max(0, start - t0), b = min(cnt, end - t0);
            for (int t = a; t < b; t++) {
                const float4 c = tile[t];
                const float dx = nx - c.x, dy = ny - c.y, dz = nz - c.z;
                const float d2 = __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
                if (d2 < top) {  // :96
                    hd[tid] = d2;
                    hi[tid] = t0 + t;
                    reheap(ns);
                    top = hd[tid];
                }
            }
        }
    }
    if (!active) return;
    for (int i = ns - 1; i > 0; i--) {  // heap_sort :40-49
        const float d0 = hd[tid];
        hd[tid] = hd[i * BS + tid];
        hd[i * BS + tid] = d0;
        const int i0 = hi[tid];
        hi[tid] = hi[i * BS + tid];
        hi[i * BS + tid] = i0;
        reheap(i);
    }
    for (int i = 0; i < ns; i++) {
        idx[(size_t)pt * ns + i] = hi[i * BS + tid];
        dist2[(size_t)pt * ns + i] = hd[i * BS + tid];
    }
}

}  // namespace p2

using namespace p2;

extern "C" {

void knnquery_cuda_launcher(int m, int nsample, const float *xyz, const float *new_xyz,
                            const int *offset, const int *new_offset, int *idx, float *dist2) {
    if (m <= 0) return;
    if (nsample < 1 || nsample > 100) { set_error("knnquery: nsample must be in [1, 100]"); return; }
    hipStream_t st = state().stream;
    const int n_total = state().total_points, nbatch = state().batch_count;
    state().total_points = 0;
    state().batch_count = 0;
    // grid-accelerated exact kNN (knn_grid.hip) when the caller lent scratch memory and announced n and b
    if (knn_grid_launch(m, nsample, n_total, nbatch, xyz, new_xyz, offset, new_offset, idx, dist2)) {
        check_launch();
        return;
    }
    auto launch = [&](auto bs_tag) {
        constexpr int BS = decltype(bs_tag)::value;
        const size_t lds = KNN_TILE * sizeof(float4) + (size_t)nsample * BS * 8;
        allow_big_lds(knn_kernel<BS>, lds);
        hipLaunchKernelGGL(knn_kernel<BS>, dim3(div_up(m, BS)), dim3(BS), lds, st, m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2);
    };
    // one-wave workgroups: more of them in flight (the per-query loop is latency-bound), small heap image
    launch(std::integral_constant<int, 64>{});
    check_launch();
}

}  // extern "C"
