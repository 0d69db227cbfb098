"""Synthetic S3DIS-like rooms: no dataset is available on the build or GPU boxes.

One room = points sampled uniformly on floor / ceiling / four walls / four boxes, one point kept per
`voxel` cell (as util/voxelize.py does for S3DIS at 0.04 m), randomly permuted, truncated to exactly
`n_points`, shifted to min 0 (util/data_util.py:197-198).
"""
import numpy as np


def _rect(rng, n, origin, u, v):
    a = rng.random((n, 1))
    b = rng.random((n, 1))
    return np.asarray(origin)[None, :] + a * np.asarray(u)[None, :] + b * np.asarray(v)[None, :]


def make_room(n_points, seed=0, voxel=0.04, dims=(7.0, 5.5, 2.8), noise=0.02, thickness=2.2):
    """-> xyz [n_points, 3] float32.  `dims` fixes the room's proportions; its size follows n_points."""
    rng = np.random.default_rng(seed)
    X, Y, Z = dims
    # surfaces: (origin, u, v)
    surf = [((0, 0, 0), (X, 0, 0), (0, Y, 0)), ((0, 0, Z), (X, 0, 0), (0, Y, 0)),
            ((0, 0, 0), (X, 0, 0), (0, 0, Z)), ((0, Y, 0), (X, 0, 0), (0, 0, Z)),
            ((0, 0, 0), (0, Y, 0), (0, 0, Z)), ((X, 0, 0), (0, Y, 0), (0, 0, Z))]
    for bx, by in ((1.0, 1.0), (4.5, 1.2), (2.0, 3.8), (5.2, 3.6)):
        s, hgt = 1.0, 0.9
        surf += [((bx, by, hgt), (s, 0, 0), (0, s, 0)),
                 ((bx, by, 0), (s, 0, 0), (0, 0, hgt)), ((bx, by + s, 0), (s, 0, 0), (0, 0, hgt)),
                 ((bx, by, 0), (0, s, 0), (0, 0, hgt)), ((bx + s, by, 0), (0, s, 0), (0, 0, hgt))]
    areas = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in surf])

    def occupied(scale):
        # oversample ~12 points per cell so almost every reachable cell is hit
        total = int(areas.sum() * scale * scale / (voxel * voxel) * 12 * thickness)
        counts = np.maximum(1, (total * areas / areas.sum()).astype(int))
        pts = np.concatenate([_rect(rng, c, np.asarray(o) * scale, np.asarray(u) * scale, np.asarray(v) * scale)
                              for c, (o, u, v) in zip(counts, surf)])
        pts += rng.normal(0, noise, pts.shape)            # scanner noise: surfaces are ~2 cells thick, off-grid
        pts = pts[rng.permutation(len(pts))]
        cell = np.floor(pts / voxel).astype(np.int64)
        cell -= cell.min(0)
        key = (cell[:, 0] * (cell[:, 1].max() + 1) + cell[:, 1]) * (cell[:, 2].max() + 1) + cell[:, 2]
        _, first = np.unique(key, return_index=True)
        return pts, first

    # the room is scaled so that its occupied-cell count just exceeds n_points (density stays that of a
    # 0.04 m voxelised scan instead of being thinned by the truncation)
    scale = np.sqrt(n_points * voxel * voxel / (areas.sum() * thickness))
    while True:
        pts, first = occupied(scale)
        if len(first) >= n_points:
            break
        scale *= max(1.02, np.sqrt(n_points / len(first)))
    keep = pts[np.sort(first)]
    keep = keep[rng.permutation(len(keep))[:n_points]]
    keep = keep - keep.min(0)
    return np.ascontiguousarray(keep, dtype=np.float32)


def make_batch(n_points_list, seed=0, voxel=0.04):
    """Several rooms concatenated the reference's way: xyz [sum n, 3], cumulative int32 offset [b]."""
    rooms = [make_room(n, seed + i, voxel) for i, n in enumerate(n_points_list)]
    return np.concatenate(rooms), np.cumsum([len(r) for r in rooms]).astype(np.int32)
