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


def data_prepare(coord, feat, label, split="train", voxel_size=0.04, voxel_max=None, rand=None, seed_index=None, feat_div=255.0):
    """The loaders' per-sample preparation on the device - util/data_util.py:181-203 (`data_prepare_v101`, bound by
    util/s3dis.py:10: feat / 255) and :206-228 (`data_prepare_scannet`, util/scannet_v2.py:10: feat_div=None) without the
    host-side transform / shuffle: shift to the minimum, one point per occupied voxel, crop to the `voxel_max` points nearest
    to a seed point ('train' splits: a random point, else the middle one), shift again.
    coord [N,3] f32 / f64, feat [N,3], label [N] on the GPU; rand / seed_index replay the loader's two np.random.randint draws
    (omitted: drawn here).  -> (coord f32, feat f32, label i64), as the loaders return them (:199-201)."""
    coord = _coord(coord)
    if voxel_size:
        coord = coord - coord.min(0)[0]
        idx = voxelize(coord, voxel_size, 0, rand)
        coord, feat, label = coord[idx], feat[idx], label[idx]
    if voxel_max and label.shape[0] > voxel_max:
        if "train" in split:
            init = int(torch.randint(0, label.shape[0], (1,))) if seed_index is None else int(seed_index)
        else:
            init = label.shape[0] // 2
        crop = crop_nearest(coord, voxel_max, init)
        coord, feat, label = coord[crop], feat[crop], label[crop]
    coord = coord - coord.min(0)[0]
    feat = feat.float()
    if feat_div:  # a true division, as torch's CPU kernels evaluate `FloatTensor / 255.` (:200): on the GPU `tensor / python scalar` multiplies
        feat = torch.div(feat, torch.tensor(float(feat_div), dtype=torch.float32, device=feat.device))  # by the fp32 reciprocal instead
    return coord.float(), feat, label.long()


# ---- readers of the scene files the loaders open (nothing is executed from the file) ----
def load_s3dis_npy(path, device="cuda"):
    """util/s3dis.py:37-41: `<room>.npy` holds one [N, 7] array xyzrgbl -> (coord [N,3], feat [N,3] rgb 0..255, label [N]) on
    `device`, in the file's dtype (np.load with allow_pickle=False)."""
    import numpy as np
    data = np.load(path, allow_pickle=False)
    if data.ndim != 2 or data.shape[1] != 7:
        raise ValueError(f"{path}: expected an [N, 7] xyzrgbl array, got {data.shape}")
    t = torch.from_numpy(np.ascontiguousarray(data)).to(device)
    return t[:, 0:3].contiguous(), t[:, 3:6].contiguous(), t[:, 6].contiguous()


def load_scannet_pth(path, device="cuda", with_label=True):
    """util/scannet_v2.py:41-47: `<scene>.pth` holds the tuple (coord [N,3], feat [N,3], label [N]) of numpy arrays that
    dataset/scannetv2/prepare_data_inst.py writes with torch.save (test split: no label).  Loaded with weights_only=True -
    the unpickler may build numpy arrays and nothing else."""
    import numpy as np
    allowed = [np.ndarray, np.dtype]
    try:
        from numpy._core.multiarray import _reconstruct
    except ImportError:  # numpy < 2
        from numpy.core.multiarray import _reconstruct
    allowed.append(_reconstruct)
    allowed += [type(np.dtype(t)) for t in (np.float32, np.float64, np.int64, np.int32, np.uint8, np.int16)]
    with torch.serialization.safe_globals(allowed):
        data = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(data, (tuple, list)) or len(data) < (3 if with_label else 2):
        raise ValueError(f"{path}: expected a (coord, feat{', label' if with_label else ''}) tuple")
    out = [torch.as_tensor(np.ascontiguousarray(np.asarray(a))).to(device) for a in data[: 3 if with_label else 2]]
    if out[0].dim() != 2 or out[0].shape[1] != 3:
        raise ValueError(f"{path}: coord must be [N, 3], got {tuple(out[0].shape)}")
    return tuple(out)
