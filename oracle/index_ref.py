"""TEST INFRASTRUCTURE — CPU restatement of the reference's index build.

Restates, with torch CPU ops where torch's own fp32 semantics matter (`//`, `round`, scalar
broadcasting) and numpy elsewhere:

  voxel_grid          torch_geometric 1.7.0 `voxel_grid` -> torch_cluster `grid_cluster`
                      (third-party, pinned by requirements.txt:13, NOT under /root/reference:
                      parity unpinned; anchored on the call site model/stratified_transformer.py:50)
  grid_sample         model/stratified_transformer.py:44-65
  get_indice_pairs    model/stratified_transformer.py:10-42   (O(M) per-window form, no [n,k,k] masks)
  csr_from_pairs      model/stratified_transformer.py:312-317 (canonical = stable sort)
  rel_pos_index       model/stratified_transformer.py:186-190
  stratified_new_offset / transition_down_offset   :283-288 / :98-102

Canonical order: `torch.argsort` (:63) and `torch.sort` (:312) are called without
stable=True in the reference; the outputs only depend on the per-query key *set*.  The oracle pins
the order a stable sort yields: per query, dense keys ascending by point index, then stratified
keys ascending by point index.
"""
import numpy as np
import torch


# ---------------------------------------------------------------------------------------------
# voxel_grid
# ---------------------------------------------------------------------------------------------
def voxel_grid(pos, batch, size, start=None, end=None):
    """pos [N,3] f32, batch [N] i64, size [3] (tensor or list), start [3] or None -> cluster [N] i64.

    torch_geometric.nn.voxel_grid (1.7.0): appends batch as a 4th coordinate with cell size 1 and
    start 0; torch_cluster.grid_cluster: start/end default to the per-dimension min/max of the
    4-column pos; voxel_d = (int64)((pos_d - start_d) / size_d) in fp32 (truncation);
    id = sum_d voxel_d * prod_{e<d} ((int64)((end_e - start_e) / size_e) + 1), x fastest.
    """
    pos = pos.float()
    size = size.tolist() if torch.is_tensor(size) else list(size)
    start = start.tolist() if torch.is_tensor(start) else start
    end = end.tolist() if torch.is_tensor(end) else end
    pos4 = torch.cat([pos, batch.unsqueeze(-1).type_as(pos)], dim=-1)
    size4 = torch.tensor(size + [1], dtype=pos.dtype)
    start4 = pos4.min(0)[0] if start is None else torch.tensor(list(start) + [0], dtype=pos.dtype)
    end4 = pos4.max(0)[0] if end is None else torch.tensor(list(end) + [int(batch.max())], dtype=pos.dtype)
    c = torch.zeros(pos.shape[0], dtype=torch.int64)
    k = 1
    for d in range(4):
        v = ((pos4[:, d] - start4[d]) / size4[d]).to(torch.int64)  # fp32 divide, truncating cast
        c += v * k
        k *= int(((end4[d] - start4[d]) / size4[d]).to(torch.int64)) + 1
    return c


def grid_sample(pos, batch, size, start):
    """model/stratified_transformer.py:44-65 -> (cluster [N] i64, p2v_map [nW,kmax] i64, counts [nW] i64)"""
    cluster = voxel_grid(pos, batch, size, start=start)
    unique, cluster, counts = torch.unique(cluster, sorted=True, return_inverse=True, return_counts=True)
    n = unique.shape[0]
    k = int(counts.max())
    p2v_map = cluster.new_zeros(n, k)
    mask = torch.arange(k).unsqueeze(0) < counts.unsqueeze(-1)
    p2v_map[mask] = torch.argsort(cluster, stable=True)
    return cluster, p2v_map, counts


# ---------------------------------------------------------------------------------------------
# pairs
# ---------------------------------------------------------------------------------------------
def window_coord(xyz, window_size, shifted):
    """:28-32.  fp32 floor division of (xyz [+ w/2] - xyz_min) by the window size tensor."""
    ws = torch.tensor([window_size] * 3).type_as(xyz)
    xyz_min = xyz.min(0)[0]
    if not shifted:
        return (xyz - xyz_min) // ws
    return (xyz + 1 / 2 * ws - xyz_min) // ws


def get_indice_pairs(p2v_map, counts, new_p2v_map, new_counts, downsample_idx, xyz, window_size, i):
    """:10-42 -> (index_0, index_1) int64, in the reference's concatenation order
    (dense pairs window-major/row-major, then stratified pairs window-major/row-major)."""
    p2v = p2v_map.numpy()
    cnt = counts.numpy()
    i0, i1 = [], []
    for w in range(p2v.shape[0]):
        pts = p2v[w, :cnt[w]]
        i0.append(np.repeat(pts, cnt[w]))
        i1.append(np.tile(pts, cnt[w]))
    N = xyz.shape[0]
    ds_mask = np.zeros(N, dtype=bool)
    ds_mask[downsample_idx.numpy().astype(np.int64)] = True
    wc = window_coord(xyz, window_size, i % 2 == 1).numpy()
    np2v = new_p2v_map.numpy()
    ncnt = new_counts.numpy()
    for w in range(np2v.shape[0]):
        pts = np2v[w, :ncnt[w]]
        key_ok = ds_mask[pts]                                  # :23,26  column mask
        c = wc[pts]                                            # [cnt,3]
        diff = (c[:, None, :] != c[None, :, :]).any(-1)        # :34
        mat = diff & key_ok[None, :]                           # :27,35  (row valid, col sampled)
        a, b = np.nonzero(mat)                                 # row-major
        i0.append(pts[a])
        i1.append(pts[b])
    return torch.from_numpy(np.concatenate(i0)), torch.from_numpy(np.concatenate(i1))


def csr_from_pairs(index_0, index_1, n_points=None):
    """:312-317 with a stable sort -> (index_0 sorted, index_1, offsets [N+1] i64, n_max int)"""
    index_0, indices = torch.sort(index_0, stable=True)
    index_1 = index_1[indices]
    counts = index_0.bincount() if n_points is None else index_0.bincount(minlength=n_points)
    n_max = int(counts.max())
    offsets = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(dim=-1)], 0)
    return index_0, index_1, offsets, n_max


def rel_pos_index(xyz, index_0, index_1, window_size, quant_size, div_mode="cuda"):
    """:186-190 -> [M,3] int32.

    div_mode: torch divides a tensor by a Python scalar as a true division on CPU but as a
    multiplication by the fp32 reciprocal on CUDA/HIP (ATen div_true_kernel_cuda, is_cpu_scalar
    branch).  The reference only ever runs on a GPU, so "cuda" is the reference's arithmetic;
    "cpu" is what importing the reference on CPU (golden fixtures) computes.
    """
    rel = xyz[index_0] - xyz[index_1]
    r = torch.round(rel * 100000)
    if div_mode == "cpu":
        rel = r / 100000
    else:
        rel = r * (torch.tensor(1.0, dtype=torch.float32) / torch.tensor(100000.0, dtype=torch.float32))
    idx = (rel + 2 * window_size - 0.0001) // quant_size
    return idx.int()


# ---------------------------------------------------------------------------------------------
# offsets (I6)
# ---------------------------------------------------------------------------------------------
def stratified_new_offset(offset, downsample_scale):
    """:283-288"""
    offset = [int(o) for o in offset]
    new_offset, count = [offset[0] // downsample_scale + 1], offset[0] // downsample_scale + 1
    for i in range(1, len(offset)):
        count += (offset[i] - offset[i - 1]) // downsample_scale + 1
        new_offset.append(count)
    return np.asarray(new_offset, dtype=np.int32)


def transition_down_offset(offset, ratio):
    """:98-102.  Note the float accumulation for b>0 (no int()); IntTensor truncates at the end."""
    offset = [int(o) for o in offset]
    n_offset, count = [int(offset[0] * ratio) + 1], int(offset[0] * ratio) + 1
    for i in range(1, len(offset)):
        count += ((offset[i] - offset[i - 1]) * ratio) + 1
        n_offset.append(count)
    return np.asarray([int(c) for c in n_offset], dtype=np.int32)


def batch_from_offset(offset):
    """:273-275"""
    offset = np.asarray(offset, dtype=np.int64)
    sizes = np.diff(np.concatenate([[0], offset]))
    return torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes)).long()


def scatter_softmax(src, index):
    """torch_scatter 2.0.6 composite/softmax.py restated with torch CPU ops (dim=0)."""
    n = int(index.max()) + 1
    idx = index.unsqueeze(-1).expand_as(src)
    mx = torch.full((n, src.shape[1]), float("-inf"), dtype=src.dtype).scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    ex = (src - mx[index]).exp()
    sm = torch.zeros((n, src.shape[1]), dtype=src.dtype).index_add_(0, index, ex)
    return ex / (sm + 1e-12)[index]


def build_stage_indices(xyz, offset, window_size, quant_size, downsample_idx, block_parity, div_mode="cuda"):
    """Everything BasicLayer.forward computes for one block of parity `block_parity` (:271-317)
    plus the rel-pos index of WindowAttention.forward (:186-188)."""
    batch = batch_from_offset(offset)
    ws = torch.tensor([window_size] * 3).type_as(xyz)
    if block_parity % 2 == 0:
        _, p2v, cnt = grid_sample(xyz, batch, ws, None)
        _, np2v, ncnt = grid_sample(xyz, batch, 2 * ws, None)
    else:
        _, p2v, cnt = grid_sample(xyz + 1 / 2 * ws, batch, ws, xyz.min(0)[0])
        _, np2v, ncnt = grid_sample(xyz + 1 / 2 * (2 * ws), batch, 2 * ws, xyz.min(0)[0])
    i0, i1 = get_indice_pairs(p2v, cnt, np2v, ncnt, downsample_idx, xyz, window_size, block_parity)
    i0, i1, offsets, n_max = csr_from_pairs(i0, i1, xyz.shape[0])
    rel = rel_pos_index(xyz, i0, i1, window_size, quant_size, div_mode)
    return dict(index_0=i0, index_1=i1, offsets=offsets, n_max=n_max, rel_idx=rel,
                p2v_map=p2v, counts=cnt, new_p2v_map=np2v, new_counts=ncnt)


# ---------------------------------------------------------------------------------------------
# Swin3D variant (model/swin3d_transformer.py): dense window pairs only, in-window quantised coordinates
# ---------------------------------------------------------------------------------------------
def swin_rel_pos_index(xyz, index_0, index_1, window_size, quant_size, shift):
    """swin3d_transformer.py:151-154 + map_func :129-130 -> [M,3] in [0, 2*int(window/quant) - 2]"""
    qgl = int(window_size / quant_size)  # :109
    xyz_quant = (xyz - xyz.min(0)[0] + shift) % window_size
    xyz_quant = xyz_quant // quant_size
    return (xyz_quant[index_0.long()] - xyz_quant[index_1.long()] + qgl - 1).int()


def swin_stage_indices(xyz, offset, window_size, quant_size, block_parity):
    """BasicLayer.forward of the Swin3D variant for one block (:239-278: every pair of points of one window; the odd
    blocks on the partition shifted by half a window), CSR with a stable sort, and the block's rel-pos index."""
    batch = batch_from_offset(offset)
    ws = torch.tensor([window_size] * 3).type_as(xyz)
    if block_parity % 2 == 0:
        _, p2v, cnt = grid_sample(xyz, batch, ws, None)
        shift = 0.0
    else:
        _, p2v, cnt = grid_sample(xyz + 1 / 2 * ws, batch, ws, xyz.min(0)[0])
        shift = 1 / 2 * ws
    n, k = p2v.shape
    mask = torch.arange(k).unsqueeze(0) < cnt.unsqueeze(-1)
    mask_mat = mask.unsqueeze(-1) & mask.unsqueeze(-2)
    i0 = p2v.unsqueeze(-1).expand(-1, -1, k)[mask_mat]
    i1 = p2v.unsqueeze(1).expand(-1, k, -1)[mask_mat]
    i0, i1, offsets, n_max = csr_from_pairs(i0, i1, xyz.shape[0])
    return dict(index_0=i0, index_1=i1, offsets=offsets, n_max=n_max, rel_idx=swin_rel_pos_index(xyz, i0, i1, window_size, quant_size, shift))


# ---------------------------------------------------------------------------------------------
# data-side step (util/voxelize.py, util/data_util.py) - numpy, stable sorts (the reference's argsort is unstable: the
# order inside a voxel / among equal distances is unspecified there; ascending index is one of its valid outputs)
# ---------------------------------------------------------------------------------------------
def fnv_hash_vec(arr):
    """util/voxelize.py:46-59"""
    import numpy as np
    arr = arr.copy().astype(np.uint64, copy=False)
    hashed = np.uint64(14695981039346656037) * np.ones(arr.shape[0], dtype=np.uint64)
    for j in range(arr.shape[1]):
        hashed *= np.uint64(1099511628211)
        hashed = np.bitwise_xor(hashed, arr[:, j])
    return hashed


def voxelize(coord, voxel_size, mode=0, rand=None):
    """util/voxelize.py:79-95 with a stable argsort and the random draw passed in"""
    import numpy as np
    discrete = np.floor(coord / np.asarray(voxel_size, dtype=coord.dtype))
    key = fnv_hash_vec(discrete)
    idx_sort = np.argsort(key, kind="stable")
    _, count = np.unique(key[idx_sort], return_counts=True)
    if mode == 0:
        sel = np.cumsum(np.insert(count, 0, 0)[0:-1]) + rand % count
        return idx_sort[sel]
    return idx_sort, count


def crop_nearest(coord, voxel_max, seed_index):
    """util/data_util.py:188-191"""
    import numpy as np
    return np.argsort(np.sum(np.square(coord - coord[seed_index]), 1), kind="stable")[:voxel_max]


def data_prepare(coord, feat, label, split="train", voxel_size=0.04, voxel_max=None, rand=None, seed_index=None, feat_div=255.0):
    """util/data_util.py:181-203 (data_prepare_v101: feat / 255, :200) and :206-228 (data_prepare_scannet: feat_div = None, :226)
    without transform / shuffle: shift to the minimum, one point per voxel (`rand` = the loader's per-voxel draw), crop to the
    voxel_max points nearest to the seed (`seed_index` = the loader's draw for 'train' splits, the middle point otherwise),
    shift again; -> (coord f32 [n,3], feat f32 [n,3], label i64 [n]).  The arithmetic stays in the coordinates' own dtype,
    as numpy 1.19.5 (requirements.txt:3) evaluates it."""
    import numpy as np
    coord = coord.copy()
    if voxel_size:
        coord -= np.min(coord, 0)
        idx = voxelize(coord, coord.dtype.type(voxel_size), 0, rand)
        coord, feat, label = coord[idx], feat[idx], label[idx]
    if voxel_max and label.shape[0] > voxel_max:
        init = int(seed_index) if "train" in split else label.shape[0] // 2
        crop = crop_nearest(coord, voxel_max, init)
        coord, feat, label = coord[crop], feat[crop], label[crop]
    coord -= np.min(coord, 0)
    feat = feat.astype(np.float32)
    return coord.astype(np.float32), feat / np.float32(feat_div) if feat_div else feat, label.astype(np.int64)
