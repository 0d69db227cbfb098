// The data-side step in front of the path: voxel subsampling keys and the crop distances of the
// reference's loaders, on the device.
//   util/voxelize.py:46-59,79-84   discrete = floor(coord / voxel);  key = FNV-style 64-bit hash of the three coordinates
//                                  (hash = 14695981039346656037; per column: hash *= 1099511628211; hash ^= column)
//   util/data_util.py:188-191      squared distance of every point to the crop's seed point (argsort + first voxel_max)
// Both in the coordinates' own precision (float32 or float64 arrays, as numpy evaluates them).
#include "common.h"

namespace p2 {

template <typename T>
__global__ void voxel_key_fnv_kernel(int N, const T *__restrict__ coord, T voxel, unsigned long long *__restrict__ keys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    unsigned long long hsh = 14695981039346656037ull;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const T q = floor(coord[(size_t)i * 3 + a] / voxel);          // np.floor(coord / voxel_size)
        const unsigned long long u = (unsigned long long)(long long)q;  // .astype(np.uint64) of a (non-negative) float
        hsh *= 1099511628211ull;
        hsh ^= u;
    }
    keys[i] = hsh;
}

template <typename T>
__global__ void crop_dist_kernel(int N, const T *__restrict__ coord, int seed, T *__restrict__ dist) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    T s = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {  // np.sum(np.square(coord - coord[init_idx]), 1): left to right
        const T d = coord[(size_t)i * 3 + a] - coord[(size_t)seed * 3 + a];
        const T sq = d * d;
        s = a == 0 ? sq : s + sq;
    }
    dist[i] = s;
}

}  // namespace p2

using namespace p2;

extern "C" {

void pointops2_voxel_keys_launcher(int N, int is_f64, const void *coord, double voxel, unsigned long long *keys) {
    if (N <= 0) return;
    if (is_f64)
        hipLaunchKernelGGL(voxel_key_fnv_kernel<double>, dim3(div_up(N, 256)), dim3(256), 0, state().stream, N, (const double *)coord, voxel, keys);
    else
        hipLaunchKernelGGL(voxel_key_fnv_kernel<float>, dim3(div_up(N, 256)), dim3(256), 0, state().stream, N, (const float *)coord, (float)voxel, keys);
    check_launch();
}

void pointops2_crop_dist_launcher(int N, int is_f64, const void *coord, int seed, void *dist) {
    if (N <= 0) return;
    if (seed < 0 || seed >= N) { set_error("pointops2_crop_dist: seed point out of range"); return; }
    if (is_f64)
        hipLaunchKernelGGL(crop_dist_kernel<double>, dim3(div_up(N, 256)), dim3(256), 0, state().stream, N, (const double *)coord, seed, (double *)dist);
    else
        hipLaunchKernelGGL(crop_dist_kernel<float>, dim3(div_up(N, 256)), dim3(256), 0, state().stream, N, (const float *)coord, seed, (float *)dist);
    check_launch();
}

}  // extern "C"
