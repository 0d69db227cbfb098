"""The data-side step immediately upstream of the path, on the device: the reference's loaders voxel-
subsample a scene and crop it to the `voxel_max` points nearest to a seed (util/voxelize.py:46-95, util/data_util.py:181-199)
with numpy on the host, per sample.  Keys and distances come from HIP kernels (csrc/dataprep.hip), the sorts are stable
device sorts: where numpy's unstable argsort leaves the order inside a voxel (or among equal distances) unspecified, the
ascending-index order is pinned - every such order is a valid output of the reference."""
import ctypes

import torch

from . import _lib
from ._lib import ptr


def _coord(coord):
    if not coord.is_cuda or coord.dim() != 2 or coord.shape[1] != 3 or coord.dtype not in (torch.float32, torch.float64):
        raise RuntimeError("expected a [N, 3] float32 / float64 GPU tensor")
    return coord.contiguous()


def voxel_keys(coord, voxel_size):
    """util/voxelize.py:79-84 (hash_type='fnv'): the 64-bit key of every point's voxel, as int64 bit patterns"""
    coord = _coord(coord)
    keys = torch.empty(coord.shape[0], dtype=torch.int64, device=coord.device)
    _lib.call("pointops2_voxel_keys_launcher", coord.shape[0], int(coord.dtype == torch.float64), ptr(coord), ctypes.c_double(float(voxel_size)), ptr(keys),
              device=coord.device)
    return keys


def voxelize(coord, voxel_size=0.05, mode=0, rand=None):
    """util/voxelize.py:79-95.  mode 0 (train): one point per occupied voxel - `rand` [n_voxels] non-negative integers stands
    for the loader's np.random.randint(0, count.max(), count.size) draw (omitted: drawn here); returns idx_unique.
    mode 1 (val): (idx_sort, count).  Voxels are ordered by their UNSIGNED 64-bit key, as numpy orders uint64."""
    keys = voxel_keys(coord, voxel_size)
    order = torch.sort(keys ^ (-2 ** 63), stable=True)[1]  # signed sort of the bias-flipped pattern = unsigned order
    _, count = torch.unique_consecutive(keys[order], return_counts=True)
    if mode != 0:
        return order, count
    start = torch.cumsum(count, 0) - count
    if rand is None:
        rand = torch.randint(0, int(count.max()), (count.shape[0],), device=coord.device)
    return order[start + rand.to(count.device) % count]


def crop_nearest(coord, voxel_max, seed_index):
    """util/data_util.py:188-191: the `voxel_max` points nearest to point `seed_index` (ascending distance, ties by index)"""
    coord = _coord(coord)
    dist = torch.empty(coord.shape[0], dtype=coord.dtype, device=coord.device)
    _lib.call("pointops2_crop_dist_launcher", coord.shape[0], int(coord.dtype == torch.float64), ptr(coord), int(seed_index), ptr(dist), device=coord.device)
    return torch.sort(dist, stable=True)[1][:voxel_max]
