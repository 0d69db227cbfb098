"""On-device index build for one Stratified Transformer stage.

Produces, bit-identically to the reference's model code in its canonical (stable-sort) order, what
`BasicLayer.forward` / `WindowAttention.forward` compute with `[nW,k,k]` boolean masks, a 12 M-element
sort per block and four host syncs (model/stratified_transformer.py:10-65, 186-190, 271-317):

    window partition (grid_sample)   -> per-point window ids, window-sorted point lists
    pair list (get_indice_pairs+CSR) -> index_1 [M] i32, index_0_offsets [N+1] i32 (+ index_0, n_max)
    relative-position index          -> rel_idx [M,3] i32

Algorithm (O(M), no masks, no pair sort): points are bucketed by window id with one stable sort per
partition; a query's dense keys are its small window's bucket, its stratified keys the sampled points
of its large window's bucket whose small-window coordinate differs from the query's; both lists are
already ascending by point index, so the CSR is written directly at prefix-summed offsets.

This module is the host-side orchestration.  It runs on whatever device the inputs live on (the
tests compare it with the oracle on CPU); on the GPU the heavy steps dispatch to HIP kernels when
`use_hip=True` (stratified_transformer_amd/csrc/index.hip).
"""
import ctypes
import os
from dataclasses import dataclass

import torch

RECIP_1E5 = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(100000.0, dtype=torch.float32)


@dataclass
class WindowPartition:
    """One grid_sample() result in segment form."""
    cluster: torch.Tensor      # [N] i64   dense window id per point (rank of the voxel id)
    order: torch.Tensor        # [N] i64   point ids sorted by (window, point id)
    starts: torch.Tensor       # [nW+1] i64 bucket boundaries into `order`
    n_windows: int


@dataclass
class BlockIndex:
    index_0: torch.Tensor          # [M] i32 (ascending)
    index_1: torch.Tensor          # [M] i32
    offsets: torch.Tensor          # [N+1] i32
    n_max: torch.Tensor            # 0-dim i64 on device, like the model's (:315)
    rel_idx: torch.Tensor          # [M,3] i32
    n_dense: torch.Tensor          # [N] i64 dense keys per query (diagnostics)
    cells: object = None           # CellPlan of the same pattern (stage_index_hip(..., cell_table_rows=L)), else None
    shard: object = None           # (QueryShard, bounds, (rank, world)) cached by pipeline.attention_block for a sharded scene
    parts: object = None           # the stage's four window partitions (stage_index_hip): sharding by window ownership needs "large"
    owner: object = None           # (owner_of, bounds, order) of the stage (sharding.window_owners), shared by both patterns
    halo: object = None            # cached (HaloShard | (plan, HaloPlan), (rank, world, kind)) of this pattern
    cells_ready: object = None     # event on the build's stream: the cell plan is complete (the pair-list tensors may still be in flight)


def batch_ids(offset, n):
    """:273-275 without the Python list comprehension: batch id per point from cumulative offsets."""
    offset = offset.to(torch.int64)
    return torch.searchsorted(offset, torch.arange(n, device=offset.device), right=True)


def voxel_ids(pos, batch, size, start):
    """torch_geometric 1.7.0 voxel_grid -> torch_cluster grid_cluster arithmetic (see compat.voxel_grid)."""
    from .compat import voxel_grid
    return voxel_grid(pos, batch, size, start=start)


def partition(pos, batch, size, start):
    """grid_sample (:44-65) in segment form; `order`/`starts` replace the zero-padded p2v_map/counts."""
    vid = voxel_ids(pos, batch, size, start)
    uniq, cluster, counts = torch.unique(vid, sorted=True, return_inverse=True, return_counts=True)
    order = torch.argsort(cluster, stable=True)
    starts = torch.zeros(counts.shape[0] + 1, dtype=torch.int64, device=pos.device)
    starts[1:] = counts.cumsum(0)
    return WindowPartition(cluster, order, starts, int(uniq.shape[0]))


def p2v_map(part):
    """Materialises the reference's zero-padded p2v_map/counts from a partition (tests only)."""
    counts = part.starts[1:] - part.starts[:-1]
    k = int(counts.max())
    out = torch.zeros(part.n_windows, k, dtype=torch.int64, device=counts.device)
    mask = torch.arange(k, device=counts.device).unsqueeze(0) < counts.unsqueeze(-1)
    out[mask] = part.order
    return out, counts


def window_coord(xyz, window_size, shifted):
    """:28-32 fp32 floor-division coordinate used by the stratified mask."""
    ws = torch.tensor([window_size] * 3).type_as(xyz)
    xyz_min = xyz.min(0)[0]
    if not shifted:
        return (xyz - xyz_min) // ws
    return (xyz + 1 / 2 * ws - xyz_min) // ws


def rel_pos_index(xyz, index_0, index_1, window_size, quant_size):
    """:186-188 with the GPU's arithmetic for `/ 100000` (ATen multiplies by the fp32 reciprocal when a
    CUDA/HIP tensor is divided by a Python scalar); evaluated identically on any device."""
    rel = xyz[index_0.long()] - xyz[index_1.long()]
    rel = torch.round(rel * 100000) * RECIP_1E5.to(xyz.device)
    return ((rel + 2 * window_size - 0.0001) // quant_size).int()


def _expand(counts):
    """row id and position-within-row for a ragged layout with `counts` entries per row."""
    rows = torch.repeat_interleave(torch.arange(counts.shape[0], device=counts.device), counts)
    starts = counts.cumsum(0) - counts
    local = torch.arange(rows.shape[0], device=counts.device) - starts[rows]
    return rows, local


def build_block_index(xyz, small, large, downsample_idx, window_size, quant_size, shifted):
    """Pair list + rel-pos index of one block (even block: shifted=False, odd: True).

    small / large: WindowPartition of the (shifted) small / 2x windows; downsample_idx [m] i32 = FPS subset.
    """
    N, dev = xyz.shape[0], xyz.device
    # dense keys: the whole bucket of the query's small window (ascending point id inside a bucket)
    s_cnt = (small.starts[1:] - small.starts[:-1])
    n_dense = s_cnt[small.cluster]
    # stratified candidates: sampled points bucketed by large window
    sampled = torch.zeros(N, dtype=torch.bool, device=dev)
    sampled[downsample_idx.long()] = True
    l_sorted_sampled = large.order[sampled[large.order]]                 # sampled point ids, (window, id) order
    l_cnt = torch.bincount(large.cluster[l_sorted_sampled], minlength=large.n_windows)
    l_starts = l_cnt.cumsum(0) - l_cnt
    cand_cnt = l_cnt[large.cluster]
    q, local = _expand(cand_cnt)
    cand = l_sorted_sampled[l_starts[large.cluster[q]] + local]
    wc = window_coord(xyz, window_size, shifted)
    keep = (wc[q] != wc[cand]).any(-1)
    sq, sk = q[keep], cand[keep]
    n_strat = torch.bincount(sq, minlength=N)
    total = n_dense + n_strat
    offsets = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    offsets[1:] = total.cumsum(0)
    M = int(offsets[-1])
    index_1 = torch.empty(M, dtype=torch.int64, device=dev)
    dq, dl = _expand(n_dense)
    index_1[offsets[dq] + dl] = small.order[small.starts[small.cluster[dq]] + dl]
    s_local = torch.arange(sq.shape[0], device=dev) - (n_strat.cumsum(0) - n_strat)[sq]
    index_1[offsets[sq] + n_dense[sq] + s_local] = sk
    index_0 = torch.repeat_interleave(torch.arange(N, device=dev), total)
    rel = rel_pos_index(xyz, index_0, index_1, window_size, quant_size)
    return BlockIndex(index_0.int(), index_1.int(), offsets.int(), total.max(), rel, n_dense)


def stratified_new_offset(offset, downsample_scale):
    """:283-288 on host integers"""
    offs = [int(o) for o in offset]
    out, count = [offs[0] // downsample_scale + 1], offs[0] // downsample_scale + 1
    for i in range(1, len(offs)):
        count += (offs[i] - offs[i - 1]) // downsample_scale + 1
        out.append(count)
    return out


def transition_down_offset(offset, ratio):
    """:98-102 (float accumulation for b>0, truncated at the end by IntTensor)"""
    offs = [int(o) for o in offset]
    out, count = [int(offs[0] * ratio) + 1], int(offs[0] * ratio) + 1
    for i in range(1, len(offs)):
        count += ((offs[i] - offs[i - 1]) * ratio) + 1
        out.append(count)
    return [int(c) for c in out]


def stage_partitions(xyz, offset, window_size):
    """The four grid_sample calls of BasicLayer.forward (:277,280,297,300)."""
    batch = batch_ids(offset, xyz.shape[0])
    ws = torch.tensor([window_size] * 3).type_as(xyz)
    xyz_min = xyz.min(0)[0]
    return {
        "small": partition(xyz, batch, ws, None),
        "small_shift": partition(xyz + 1 / 2 * ws, batch, ws, xyz_min),
        "large": partition(xyz, batch, 2 * ws, None),
        "large_shift": partition(xyz + 1 / 2 * (2 * ws), batch, 2 * ws, xyz_min),
    }


# ---------------------------------------------------------------------------------------------
# native path (csrc/index.hip): same outputs, HIP kernels, two host syncs per stage
# ---------------------------------------------------------------------------------------------
@dataclass
class HipPartition:
    cluster: torch.Tensor    # [N] i32
    order: torch.Tensor      # [N] i32
    starts: torch.Tensor     # [N+2] i32
    n_windows: torch.Tensor  # [1] i32 (device)


def _f32(x):
    import numpy as np
    return float(np.float32(x))


class CellPlanStruct(ctypes.Structure):
    """pointops2_cell_plan of include/pointops2_hip.h"""
    _fields_ = [("n_points", ctypes.c_int), ("n_cells", ctypes.c_int), ("n_parents", ctypes.c_int), ("n_pairs", ctypes.c_int),
                ("n_keyslots", ctypes.c_int), ("counts", ctypes.c_void_p), ("parent_first", ctypes.c_void_p), ("cell_perm", ctypes.c_void_p), ("cell_qstart", ctypes.c_void_p),
                ("cell_kbase", ctypes.c_void_p), ("cell_pbase", ctypes.c_void_p), ("cell_order", ctypes.c_void_p),
                ("qcell", ctypes.c_void_p), ("cell_keys", ctypes.c_void_p), ("kcell", ctypes.c_void_p), ("relp", ctypes.c_void_p),
                ("task_first", ctypes.c_int), ("task_step", ctypes.c_int), ("table_rows", ctypes.c_int), ("max_queries", ctypes.c_int),
                ("task_list", ctypes.c_void_p), ("task_count", ctypes.c_void_p)]


@dataclass
class CellPlan:
    """The window-centric view of one block pattern (csrc/index.hip "cells", csrc/cell_attn.hip): queries grouped by
    (small window, large window); a cell's queries share one candidate key list, so a cell is a dense n_q x n_k tile."""
    n_points: int
    n_cells: int
    n_pairs: int                   # P = sum n_q * n_k (tile entries, incl. the flagged non-keys)
    n_keyslots: int                # K = sum n_k
    nk_max: int
    table_rows: int                # L the packed rel-pos indices were clamped to
    n_parents: int                 # cells before the cut into pieces of at most cell_max_queries queries
    counts: torch.Tensor           # [8] i32
    parent_first: torch.Tensor     # [N+2] i32
    cell_perm: torch.Tensor        # [N] i32
    cell_desc: torch.Tensor        # [N,4] i32 {dense start, dense count, candidate start, candidate count}
    cell_qstart: torch.Tensor      # [N+2] i32
    cell_kbase: torch.Tensor       # [N+2] i32
    cell_pbase: torch.Tensor       # [N+2] i32
    cell_order: torch.Tensor       # [N] i32
    qcell: torch.Tensor            # [N] i32
    cell_keys: torch.Tensor        # [K] i32
    kcell: torch.Tensor            # [K] i32
    relp: torch.Tensor             # [P] i32 (bit pattern of the packed word)
    struct: CellPlanStruct = None
    max_queries: int = 0           # the cut of pass 1 (cell_max_queries; 0 = uncut cells)
    task_list: torch.Tensor = None   # optional [<= n_cells] i32 cell ids + task_count [1] i32: an explicit share (with_tasks())
    task_count: torch.Tensor = None
    task_first: int = 0            # this launch's share of the cells: cell_perm[task_first::task_step] (share(): one scene over ranks)
    task_step: int = 1

    def c_arg(self):
        from ._lib import ptr
        if self.struct is None:
            self.struct = CellPlanStruct(self.n_points, self.n_cells, self.n_parents, self.n_pairs, self.n_keyslots, ptr(self.counts),
                                         ptr(self.parent_first), ptr(self.cell_perm), ptr(self.cell_qstart),
                                         ptr(self.cell_kbase), ptr(self.cell_pbase), ptr(self.cell_order), ptr(self.qcell), ptr(self.cell_keys),
                                         ptr(self.kcell), ptr(self.relp), int(self.task_first), int(self.task_step), int(self.table_rows), int(self.max_queries),
                                         ptr(self.task_list), ptr(self.task_count))
        return ctypes.byref(self.struct)

    @property
    def partial(self):
        return self.task_step > 1 or self.task_list is not None

    def with_tasks(self, task_list, task_count):
        """The same plan restricted to the cells `task_list[:task_count[0]]` (cell ids, device tensors): sharding.py assigns cells
        to the rank that owns their first query."""
        import dataclasses
        return dataclasses.replace(self, struct=None, task_first=0, task_step=1, task_list=task_list, task_count=task_count)

    def share(self, rank, world):
        """The same plan restricted to every world-th cell (by size order) starting at `rank`: the unit of sharding.sharded_cell_attention."""
        import dataclasses
        return dataclasses.replace(self, struct=None, task_first=int(rank), task_step=int(world))

    def tensors(self):
        return (self.counts, self.parent_first, self.cell_perm, self.cell_desc, self.cell_qstart, self.cell_kbase, self.cell_pbase, self.cell_order, self.qcell,
                self.cell_keys, self.kcell, self.relp)


def cell_query_cap(n_points, heads):
    """Queries per cell piece: small enough that cells x heads fill the chip's resident waves several times over."""
    # measured on the four stages of the S3DIS configuration (points x heads = 300k, 150k, 75k, 37k; tools/bench_cell.py): pieces of 32
    # queries for the two large ones, 16 for the two small ones (8 / 4 there cost 5-25 % of the backward: every piece flushes its
    # keys' gradients; 32 there leaves too few pieces for the chip)
    big, small = int(os.environ.get("P2_CELL_CAP_BIG", 32)), int(os.environ.get("P2_CELL_CAP_SMALL", 16))   # (experiment knobs)
    return big if n_points * heads >= 96000 else small


class _KeyOverflow(Exception):
    """a voxel coordinate did not fit the fixed-width key of the one-sort partitions"""


FUSED_PARTITIONS = os.environ.get("P2_PARTITIONS4", "1") != "0"


def stage_partitions_hip(xyz, offset, window_size, one_sort=None, cell_max_queries=None):
    """The part of a stage's index build that needs the coordinates only: bounding box and the four window partitions
    (grid_sample x 4, stratified_transformer.py:277,280,297,300).  Returns the context stage_index_hip continues from - a caller
    can run this beside the stage's FPS instead of behind it.

    one_sort (default): all four partitions by ONE radix sort on a key of fixed width (csrc/index.hip, voxel_key4_kernel) and no
    host sync; `overflow` (a device flag the next read-back of stage_index_hip looks at) says that a voxel coordinate needed more
    than ten bits - stage_index_hip then comes back here with one_sort=False: one sort per partition, the key sized from the
    bounding box on the host (one sync), any extent.

    cell_max_queries (>= 0; None = no cell plans wanted): also the half of the cell plans' first pass that needs the partitions only
    (pointops2_cell_plan_prepare_launcher: the cells, their cut into pieces, the parents - ~22 of the ~38 launches of that pass), so
    that a caller who runs this beside the stage's sampler has them off the path between the sampler's end and the first block."""
    import numpy as np
    from . import _lib
    from ._lib import ptr
    assert xyz.is_cuda and xyz.dtype == torch.float32 and xyz.is_contiguous()
    N, b, dev = xyz.shape[0], offset.shape[0], xyz.device
    l = _lib.lib()
    w32 = np.float32(window_size)
    i32 = dict(dtype=torch.int32, device=dev)
    if one_sort is None:
        one_sort = FUSED_PARTITIONS
    names = ("small", "small_shift", "large", "large_shift")
    with torch.cuda.device(dev):
        ws_bytes = int(l.pointops2_index_workspace_bytes(N))
        bbox = torch.empty(6, dtype=torch.float32, device=dev)
        _lib.call("pointops2_bbox_launcher", N, ptr(xyz), ptr(bbox), device=dev)
        parts, overflow = {}, None
        if one_sort and 0 < N < 2 ** 28:
            ws_bytes = max(ws_bytes, int(l.pointops2_partitions4_workspace_bytes(N)))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            cluster, order, starts = torch.empty((4, N), **i32), torch.empty((4, N), **i32), torch.empty((4, N + 2), **i32)
            n_windows, overflow = torch.empty(4, **i32), torch.empty(1, **i32)
            _lib.call("pointops2_window_partitions4_launcher", N, b, ptr(xyz), ptr(offset), ptr(bbox), float(w32), ptr(cluster), ptr(order),
                      ptr(starts), ptr(n_windows), ptr(overflow), ptr(ws), ws_bytes, device=dev)
            for v, name in enumerate(names):
                parts[name] = HipPartition(cluster[v], order[v], starts[v], n_windows[v:v + 1])
        else:
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            bb = np.asarray(bbox.tolist(), dtype=np.float32)   # host sync: sizes the radix-sort key
            for name, size, shift in ((names[0], w32, np.float32(0)), (names[1], w32, np.float32(0.5) * w32),
                                      (names[2], np.float32(2) * w32, np.float32(0)), (names[3], np.float32(2) * w32, w32)):
                nvox = 1
                for a in range(3):
                    nvox *= int((np.float32(bb[3 + a] + shift) - bb[a]) / size) + 1
                key_bits = max(int(nvox * b).bit_length() + 1, 8)
                part = HipPartition(torch.empty(N, **i32), torch.empty(N, **i32), torch.empty(N + 2, **i32), torch.empty(1, **i32))
                _lib.call("pointops2_window_partition_launcher", N, b, ptr(xyz), ptr(offset), ptr(bbox), float(size), float(shift), key_bits,
                          ptr(part.cluster), ptr(part.order), ptr(part.starts), ptr(part.n_windows), ptr(ws), ws_bytes, device=dev)
                parts[name] = part
        prepared = None
        if cell_max_queries is not None and N > 0:
            prepared = {"cap": int(cell_max_queries)}
            cws_bytes = int(l.pointops2_cell_plan_workspace_bytes(N))
            for shifted, sname, lname in ((0, "small", "large"), (1, "small_shift", "large_shift")):
                cells = _new_cells(N, i32)
                cws = torch.empty(cws_bytes, dtype=torch.uint8, device=dev)   # (its own: it carries the pass's state to stage_index_hip)
                _lib.call("pointops2_cell_plan_prepare_launcher", N, int(cell_max_queries), ptr(parts[sname].cluster), ptr(parts[lname].cluster),
                          ptr(cells["cell_order"]), ptr(cells["parent_first"]), ptr(cells["counts"]), ptr(cws), cws_bytes, device=dev)
                prepared[shifted] = (cells, cws, cws_bytes)
    return dict(parts=parts, ws=ws, ws_bytes=ws_bytes, bbox=bbox, w32=w32, overflow=overflow, cells_prepared=prepared)


def _new_cells(N, i32):
    return dict(counts=torch.empty(8, **i32), parent_first=torch.empty(N + 2, **i32), cell_perm=torch.empty(N, **i32), cell_desc=torch.empty((N, 4), **i32),
                cell_qstart=torch.empty(N + 2, **i32), cell_kbase=torch.empty(N + 2, **i32), cell_pbase=torch.empty(N + 2, **i32),
                cell_order=torch.empty(N, **i32), qcell=torch.empty(N, **i32))


def stage_index_hip(xyz, offset, window_size, quant_size, downsample_idx, cell_table_rows=None, cell_max_queries=0, partitions=None,
                    on_even=None, patterns=(0, 1)):
    """Even and odd block index of one stage, built by the HIP kernels of csrc/index.hip.

    xyz [N,3] f32 (GPU), offset [b] i32, downsample_idx [m] i32 -> (BlockIndex even, BlockIndex odd),
    bit-identical to build_block_index() on the same inputs.  cell_table_rows = L: also the cell plan of both
    patterns (BlockIndex.cells) for fused.cell_attention, with the rel-pos indices clamped to [0, L); cell_max_queries > 0
    cuts cells into pieces of at most that many queries (cell_query_cap).  partitions: the result of stage_partitions_hip
    on the same xyz / offset / window_size, when the caller has already run it.  patterns=(0,) / (1,): only the plain / only the
    shifted pattern (the other BlockIndex is None) - what ONE block of the unmodified model needs (:302-317 rebuilds per block)."""
    assert xyz.is_cuda and xyz.dtype == torch.float32 and xyz.is_contiguous()
    ctx = partitions if partitions is not None else stage_partitions_hip(xyz, offset, window_size)
    try:
        return _stage_index_hip(ctx, xyz, offset, quant_size, downsample_idx, cell_table_rows, cell_max_queries, on_even, patterns)
    except _KeyOverflow:
        # the scene spans more than 1024 windows along an axis: partitions with host-sized keys, and the build again (the caller's
        # context is corrected in place - pipeline.scene_pass keeps it for the per-block rebuilds of the model-order leg)
        ctx.update(stage_partitions_hip(xyz, offset, window_size, one_sort=False))
        return _stage_index_hip(ctx, xyz, offset, quant_size, downsample_idx, cell_table_rows, cell_max_queries, on_even, patterns)


def _stage_index_hip(ctx, xyz, offset, quant_size, downsample_idx, cell_table_rows, cell_max_queries, on_even, patterns):
    import numpy as np
    from . import _lib
    from ._lib import ptr
    N, b, dev = xyz.shape[0], offset.shape[0], xyz.device
    m = int(downsample_idx.shape[0])
    l = _lib.lib()
    i32 = dict(dtype=torch.int32, device=dev)

    def call(name, *args):
        _lib.call(name, *args, device=dev)

    parts, ws, ws_bytes, bbox, w32 = ctx["parts"], ctx["ws"], ctx["ws_bytes"], ctx["bbox"], ctx["w32"]
    overflow = ctx.get("overflow")
    with torch.cuda.device(dev):
        sampled = torch.zeros(N, **i32)
        out = [None, None]   # [plain, shifted]
        pending = []

        def finish(pend):
            """host sync: M of the patterns in `pend` (and their cell plans' totals), then the fills"""
            flag = [] if overflow is None or ctx.get("overflow_checked") else [overflow]
            if cell_table_rows is None:
                flat = torch.cat([p[5][N:N + 1] for p in pend] + flag).tolist()
                totals = [(flat[i], None) for i in range(len(pend))]
            else:
                flat = torch.cat([torch.cat([p[5][N:N + 1], p[6]["counts"]]) for p in pend] + flag).tolist()
                totals = [(flat[9 * i], flat[9 * i + 1:9 * i + 6]) for i in range(len(pend))]
            if flag:
                if flat[-1]:
                    raise _KeyOverflow()
                ctx["overflow_checked"] = True
            # The cell plans first: the window-centric kernels need nothing else of a pattern, so a caller that waits for `cells_ready`
            # (pipeline.scene_pass) starts its attention blocks while the pair lists - 100 us per pattern at stage 0 - are still being written.
            fills = []
            for (s, lg, ls, ls_starts, wc, offsets, cells, which), (M, ccounts) in zip(pend, totals):
                index_0, index_1 = torch.empty(M, **i32), torch.empty(M, **i32)
                rel = torch.empty((M, 3), **i32)
                counts = offsets[1:] - offsets[:-1]
                plan = None
                if cells is not None:
                    n_cells, P, K, nk_max, n_parents = ccounts
                    if P < 0 or P >= 2 ** 31 - 1:
                        raise RuntimeError("cell plan: more than 2^31 tile entries")
                    cell_keys, kcell, relp = torch.empty(max(K, 1), **i32), torch.empty(max(K, 1), **i32), torch.empty(max(P, 1), **i32)
                    call("pointops2_cell_plan_fill_launcher", N, ptr(xyz), float(w32), _f32(quant_size), int(cell_table_rows), ptr(s.order), ptr(ls),
                         ptr(wc), ptr(cells["cell_order"]), ptr(cells["qcell"]), ptr(cells["cell_qstart"]), ptr(cells["cell_desc"]),
                         ptr(cells["cell_kbase"]), ptr(cells["cell_pbase"]), ptr(cell_keys), ptr(kcell), ptr(relp))
                    plan = CellPlan(N, n_cells, P, K, nk_max, int(cell_table_rows), n_parents, cell_keys=cell_keys, kcell=kcell, relp=relp,
                                    max_queries=int(cell_max_queries), **cells)
                out[which] = BlockIndex(index_0, index_1, offsets, counts.max(), rel, None, plan, parts=parts)
                # the small-window partition's point order groups the rows of the pair list by window: the operators' pair walkers
                # take their rows in it (pointops.row_order_of) without sorting anything
                from . import pointops as _P
                _P.seed_row_order(offsets, index_1, s.order, out[which].n_max)
                fills.append((s, lg, ls, ls_starts, wc, offsets, index_0, index_1, rel))
            if cell_table_rows is not None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                for (_, _, _, _, _, _, _, which), _t in zip(pend, totals):
                    out[which].cells_ready = ev
            for s, lg, ls, ls_starts, wc, offsets, index_0, index_1, rel in fills:
                call("pointops2_pairs_fill_launcher", N, ptr(xyz), float(w32), _f32(quant_size), ptr(s.cluster), ptr(s.order), ptr(s.starts),
                     ptr(lg.cluster), ptr(ls), ptr(ls_starts), ptr(wc), ptr(offsets), ptr(index_0), ptr(index_1), ptr(rel))

        for shifted, sname, lname in ((0, "small", "large"), (1, "small_shift", "large_shift")):
            if shifted not in patterns:
                continue
            s, lg = parts[sname], parts[lname]
            ls, ls_starts = torch.empty(max(m, 1), **i32), torch.empty(N + 1, **i32)
            call("pointops2_sampled_buckets_launcher", N, m, ptr(downsample_idx), ptr(lg.order), ptr(lg.starts), ptr(lg.n_windows),
                 ptr(sampled), ptr(ls), ptr(ls_starts), ptr(ws), ws_bytes)
            wc = torch.empty((N, 3), dtype=torch.float32, device=dev)
            call("pointops2_window_coord_launcher", N, ptr(xyz), ptr(bbox), float(w32), shifted, ptr(wc))
            offsets = torch.empty(N + 1, **i32)
            call("pointops2_pairs_count_launcher", N, ptr(s.cluster), ptr(s.starts), ptr(lg.cluster), ptr(ls), ptr(ls_starts), ptr(wc),
                 ptr(offsets), ptr(ws), ws_bytes)
            cells = None
            if cell_table_rows is not None:
                ready = ctx.get("cells_prepared")
                if ready is not None and ready.get("cap") == int(cell_max_queries) and shifted in ready and not ctx.get("cells_prepared_used", {}).get(shifted):
                    # the first half of the pass ran with the partitions (stage_partitions_hip): only what needs the samples is left
                    cells, cws, cws_bytes = ready[shifted]
                    ctx.setdefault("cells_prepared_used", {})[shifted] = True   # (its arrays become this build's plan: not handed out twice)
                    call("pointops2_cell_plan_sizes_launcher", N, ptr(s.cluster), ptr(s.starts), ptr(lg.cluster), ptr(ls_starts), ptr(cells["cell_order"]),
                         ptr(cells["qcell"]), ptr(cells["cell_desc"]), ptr(cells["cell_qstart"]), ptr(cells["cell_kbase"]), ptr(cells["cell_pbase"]),
                         ptr(cells["cell_perm"]), ptr(cells["counts"]), ptr(cws), cws_bytes)
                else:
                    cws_bytes = int(l.pointops2_cell_plan_workspace_bytes(N))
                    cws = ws if cws_bytes <= ws_bytes else torch.empty(cws_bytes, dtype=torch.uint8, device=dev)
                    cells = _new_cells(N, i32)
                    call("pointops2_cell_plan_count_launcher", N, int(cell_max_queries), ptr(s.cluster), ptr(s.starts), ptr(lg.cluster), ptr(ls_starts),
                         ptr(cells["cell_order"]), ptr(cells["qcell"]), ptr(cells["cell_desc"]), ptr(cells["cell_qstart"]), ptr(cells["cell_kbase"]),
                         ptr(cells["cell_pbase"]), ptr(cells["cell_perm"]), ptr(cells["parent_first"]), ptr(cells["counts"]), ptr(cws),
                         max(cws_bytes, ws_bytes))
            pending.append((s, lg, ls, ls_starts, wc, offsets, cells, shifted))
            if on_even is not None and shifted == 0:
                # the caller wants the plain pattern as soon as it exists (its first block runs beside the shifted pattern's
                # build): one more host sync, the plain pattern ~0.4 ms earlier
                finish(pending)
                pending = []
                on_even(out[0])
        if pending:
            finish(pending)
    return out[0], out[1], parts


# ---------------------------------------------------------------------------------------------
# Swin3D variant (model/swin3d_transformer.py): the same three operators on dense window pairs, tables of
# 2*int(window/quant) - 1 rows, rel-pos index = difference of the points' quantised in-window coordinates
# ---------------------------------------------------------------------------------------------
def swin_table_rows(window_size, quant_size):
    return 2 * int(window_size / quant_size) - 1  # swin3d_transformer.py:109-117


def swin_rel_pos_index(xyz, index_0, index_1, window_size, quant_size, shift):
    """swin3d_transformer.py:151-154 + map_func :129-130 with torch ops on the tensors' device (what the model file itself runs)"""
    qgl = int(window_size / quant_size)
    xyz_quant = (xyz - xyz.min(0)[0] + shift) % window_size
    xyz_quant = xyz_quant // quant_size
    return (xyz_quant[index_0.long()] - xyz_quant[index_1.long()] + qgl - 1).int()


def swin_stage_index_hip(xyz, offset, window_size, quant_size):
    """Plain and shifted block index of one Swin3D stage (swin3d_transformer.py:239-278) from the HIP index build: the dense
    pairs of the small-window partitions (a Stratified stage without sampled keys), then the Swin rel-pos index."""
    none = torch.empty(0, dtype=torch.int32, device=xyz.device)
    even, odd, parts = stage_index_hip(xyz, offset, window_size, quant_size, none)
    ws = torch.tensor([window_size] * 3).type_as(xyz)
    for blk, shift in ((even, 0.0), (odd, 1 / 2 * ws)):
        blk.rel_idx = swin_rel_pos_index(xyz, blk.index_0, blk.index_1, window_size, quant_size, shift).contiguous()
    return even, odd, parts
