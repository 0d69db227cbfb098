"""Golden vectors of the data-side step in front of the path, produced by EXECUTING THE REFERENCE'S OWN
numpy code (build container only; nothing of the reference is copied):

  util/voxelize.py:46-95            fnv_hash_vec, voxelize (train mode 0 and val mode 1)
  util/data_util.py:181-203         data_prepare_v101 (the loader util/s3dis.py:10 binds; 'val' and 'train' splits)
  util/data_util.py:206-228         data_prepare_scannet (the loader util/scannet_v2.py:10 binds)

Shims (no reference code): `collections.Sequence` (gone in Python 3.10; voxelize.py:2 imports it), `torch_geometric.nn.
voxel_grid` (voxelize.py:4, not used by the functions executed here) and an empty `SharedArray` (data_util.py:3).
Two things are pinned while the reference runs, both a valid output of the reference:
  * its `np.argsort` calls (voxelize.py:86, data_util.py:190) are unstable (quicksort): the order inside a voxel / among equal
    distances is unspecified there -> executed with kind="stable" (ascending index);
  * its `np.random.randint` draws (voxelize.py:90: one draw per voxel; data_util.py:189: the crop's seed point) are RECORDED
    and stored in the fixture, so that the build can replay them.

float32 coordinates and numpy versions: voxelize.py:80 divides by `np.array(voxel_size)`, a 0-d float64 array.  Under the
reference's pinned numpy 1.19.5 (requirements.txt:3; value-based casting) a float32 array divided by it stays float32 with
0.04 rounded to float32; under this container's numpy 2.2 (NEP 50) the same expression is evaluated in float64.  The f32 cases
therefore call the reference with voxel_size = np.float32(0.04), which gives the pinned numpy's result under BOTH versions.

    python tests/golden/make_golden_dataprep.py   ->  tests/golden/voxelize_crop.npz
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def install_shims():
    collections.Sequence = collections.abc.Sequence
    for name, attrs in (("torch_geometric", {}), ("torch_geometric.nn", {"voxel_grid": None}), ("SharedArray", {})):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m


class Pinned:
    """stable argsort + recorded randint while the reference executes"""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._argsort, self._randint = np.argsort, np.random.randint
        np.argsort = lambda a, *args, **kw: self._argsort(a, *args, **{**kw, "kind": "stable"})

        def randint(*args, **kw):
            r = self._randint(*args, **kw)
            self.draws.append(np.asarray(r).copy())
            return r
        np.random.randint = randint
        return self

    def __exit__(self, *exc):
        np.argsort, np.random.randint = self._argsort, self._randint


def scene(n, seed, dtype):
    """a small scanned-room-like cloud with several points per 0.04 m voxel: floor + two walls + clutter, min at 0"""
    rng = np.random.default_rng(seed)
    u = rng.random((n, 3)) * np.array([2.4, 1.8, 1.2])
    which = rng.integers(0, 4, n)
    u[which == 0, 2] = 0.0
    u[which == 1, 0] = 0.0
    u[which == 2, 1] = 0.0
    u += rng.normal(0, 0.004, (n, 3))
    u -= u.min(0)
    return np.ascontiguousarray(u.astype(dtype))


def main():
    install_shims()
    sys.path.insert(0, REF)
    from util.voxelize import fnv_hash_vec, voxelize      # the reference
    from util import data_util                            # the reference
    out = {}
    for tag, dtype in (("f64", np.float64), ("f32", np.float32)):
        coord = scene(30000, 3 if tag == "f64" else 4, dtype)
        voxel = 0.04 if tag == "f64" else np.float32(0.04)   # (see the header: float32 under numpy 2)
        rng = np.random.default_rng(9)
        feat = rng.integers(0, 256, (coord.shape[0], 3)).astype(dtype)       # rgb 0..255 as the .npy rows hold it
        label = rng.integers(0, 13, coord.shape[0]).astype(dtype)
        out[f"{tag}_coord"], out[f"{tag}_feat"], out[f"{tag}_label"] = coord.copy(), feat.astype(np.uint8), label.astype(np.uint8)
        # keys of the voxels (voxelize.py:79-84)
        out[f"{tag}_keys"] = fnv_hash_vec(np.floor(coord / np.array(voxel)))
        np.random.seed(5)
        with Pinned() as pin:
            idx_unique = voxelize(coord, voxel, mode=0)
        out[f"{tag}_train_idx"], out[f"{tag}_train_rand"] = idx_unique.astype(np.int32), pin.draws[0].astype(np.int32)
        with Pinned():
            idx_sort, count = voxelize(coord, voxel, mode=1)
        out[f"{tag}_val_idx_sort"], out[f"{tag}_val_count"] = idx_sort.astype(np.int32), count.astype(np.int32)
        print(tag, "points", coord.shape[0], "voxels", count.size, "largest voxel", int(count.max()))
        for fn_name in ("data_prepare_v101", "data_prepare_scannet"):
            fn = getattr(data_util, fn_name)
            for split in ("val", "train"):
                np.random.seed(6)
                with Pinned() as pin:
                    c, f, l = fn(coord.copy(), feat.copy(), label.copy(), split=split, voxel_size=voxel, voxel_max=4000)
                key = f"{tag}_{fn_name}_{split}"
                out[key + "_coord"], out[key + "_feat"], out[key + "_label"] = c.numpy(), f.numpy(), l.numpy().astype(np.int16)
                out[key + "_rand"] = pin.draws[0].astype(np.int32)
                if split == "train":
                    out[key + "_seed"] = np.int32(pin.draws[1])
                assert c.shape[0] == 4000
    path = os.path.join(HERE, "voxelize_crop.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB")


if __name__ == "__main__":
    main()
