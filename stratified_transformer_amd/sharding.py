"""Window attention of ONE scene sharded over the GPUs of a node.

The reference has no such path (its only parallelism is DDP replicas); parity target: the sharded
result equals the single-GPU result (integers bit-identical, fp32 within 1e-3).

Scheme — shard by QUERY, never by key ownership (shifted small windows straddle shifted large windows,
so a query's keys can live in any neighbouring window):
  * every rank holds the full index of the block (the index build is replicated: it is integer work of a
    few ms, and FPS cannot be sharded bit-exactly anyway) and owns a contiguous range of queries chosen
    so that the PAIR counts (not the point counts) are balanced;
  * a rank owns the q/k/v rows of its own query range.  Forward: k and v rows are all-gathered (RCCL over
    xGMI; each rank's slice travels directly to its 7 peers), then A1..A4 run locally on the rank's CSR
    slice.  Backward: the key-side gradients (grad_k, grad_v) are reduce-scattered back to their owners
    and the three table gradients all-reduced.
  * one process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).

The local compute goes through an `ops` namespace with the operator API of `pointops` (the HIP ops by
default).  The CPU tests inject a reference implementation there; nothing in this module depends on it.
"""
from dataclasses import dataclass

import torch
import torch.distributed as dist


_TIMER = None
BYTES_MOVED = 0   # payload bytes this rank handed to collectives since reset_bytes() (bench.py --shard: collective_bytes_per_step)


def reset_bytes():
    global BYTES_MOVED
    BYTES_MOVED = 0


def _count(t):
    global BYTES_MOVED
    BYTES_MOVED += int(t.numel()) * t.element_size()


def set_timer(timer):
    """pipeline.Timer (or None): the collectives below are then recorded as comm/* spans on the current stream"""
    global _TIMER
    _TIMER = timer


def _comm(name, fn, *a, **k):
    if _TIMER is None:
        return fn(*a, **k)
    return _TIMER.run("comm/" + name, fn, *a, **k)


@dataclass
class QueryShard:
    lo: int                    # first owned query
    hi: int                    # one past the last owned query
    offsets: torch.Tensor      # [hi-lo+1] i32, rebased to 0
    index_1: torch.Tensor      # [M_local] i32 (global key ids)
    rel_idx: torch.Tensor      # [M_local,3] i32
    pair_lo: int
    pair_hi: int


def balanced_bounds(offsets, world):
    """Query boundaries [world+1] such that every rank gets ~M/world pairs (contiguous ranges)."""
    offs = offsets.to(torch.int64)
    M = int(offs[-1])
    N = offs.shape[0] - 1
    targets = torch.arange(1, world, device=offs.device, dtype=torch.int64) * M // world
    cuts = torch.searchsorted(offs, targets, right=False).clamp_(0, N)
    return [0] + [int(c) for c in cuts.tolist()] + [N]


def make_shard(block, rank, world, bounds=None):
    """Slice of a BlockIndex (index_build) owned by `rank`."""
    bounds = bounds or balanced_bounds(block.offsets, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    p0, p1 = int(block.offsets[lo]), int(block.offsets[hi])
    return QueryShard(lo, hi, (block.offsets[lo:hi + 1] - p0).to(torch.int32).contiguous(),
                      block.index_1[p0:p1].contiguous(), block.rel_idx[p0:p1].contiguous(), p0, p1), bounds


def _host_staged(group):
    """gloo moves host memory: device tensors are staged through the host (tests / rehearsals on one GPU; the product
    backend is "nccl" = RCCL, which takes device tensors directly)"""
    return dist.get_backend(group) == "gloo"


def _all_reduce_sum(t, group):
    _count(t)
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        _comm("all_reduce", dist.all_reduce, h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        _comm("all_reduce", dist.all_reduce, t, op=dist.ReduceOp.SUM, group=group)
    return t


class _GatherRows(torch.autograd.Function):
    """forward: all-gather row shards (padded to the longest shard, so every backend's fixed-size
    collective applies) into the full [N, ...] tensor; backward: reduce-scatter of the full gradient back
    to the owners' row ranges (an all-reduce + slice where the backend has no reduce-scatter, i.e. gloo)."""

    @staticmethod
    def forward(ctx, local, bounds, rank, group):
        world = len(bounds) - 1
        ctx.bounds, ctx.rank, ctx.group = bounds, rank, group
        sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
        mx = max(sizes)
        tail = tuple(local.shape[1:])
        staged = local.is_cuda and _host_staged(group)
        dev = torch.device("cpu") if staged else local.device
        padded = torch.zeros((mx,) + tail, dtype=local.dtype, device=dev)
        padded[: sizes[rank]] = local
        gathered = torch.empty((world, mx) + tail, dtype=local.dtype, device=dev)
        _count(padded)
        _comm("all_gather", dist.all_gather_into_tensor, gathered.view((world * mx,) + tail), padded, group=group)
        return torch.cat([gathered[r, : sizes[r]] for r in range(world)], 0).to(local.device)

    @staticmethod
    def backward(ctx, grad_full):
        bounds, rank, group = ctx.bounds, ctx.rank, ctx.group
        world = len(bounds) - 1
        sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
        if grad_full.is_cuda and not _host_staged(group):
            mx = max(sizes)
            tail = tuple(grad_full.shape[1:])
            padded = torch.zeros((world, mx) + tail, dtype=grad_full.dtype, device=grad_full.device)
            for r in range(world):
                padded[r, : sizes[r]] = grad_full[bounds[r]:bounds[r + 1]]
            out = torch.empty((mx,) + tail, dtype=grad_full.dtype, device=grad_full.device)
            _count(padded)
            _comm("reduce_scatter", dist.reduce_scatter_tensor, out, padded.view((world * mx,) + tail), op=dist.ReduceOp.SUM, group=group)
            return out[: sizes[rank]].contiguous(), None, None, None
        g = _all_reduce_sum(grad_full.contiguous().clone(), group)
        return g[bounds[rank]:bounds[rank + 1]].contiguous(), None, None, None


class _SharedParam(torch.autograd.Function):
    """replicated parameter: identity forward, all-reduce of the gradient backward"""

    @staticmethod
    def forward(ctx, p, group):
        ctx.group = group
        return p.view_as(p)

    @staticmethod
    def backward(ctx, g):
        return _all_reduce_sum(g.contiguous().clone(), ctx.group), None


class _Ctx:
    pass


class _ScatterSumRows(torch.autograd.Function):
    """[N, ...] partial sums on every rank -> this rank's rows [lo:hi) of their sum (reduce-scatter); backward: the ranks' row
    gradients gathered into the full tensor (all-gather).  The mirror image of _GatherRows, built from its two halves."""

    @staticmethod
    def forward(ctx, full_partial, bounds, rank, group):
        ctx.bounds, ctx.rank, ctx.group = bounds, rank, group
        c = _Ctx()
        c.bounds, c.rank, c.group = bounds, rank, group
        return _GatherRows.backward(c, full_partial.contiguous())[0]

    @staticmethod
    def backward(ctx, g_local):
        return _GatherRows.forward(_Ctx(), g_local.contiguous(), ctx.bounds, ctx.rank, ctx.group), None, None, None


def sharded_cell_attention(cell_attention, plan, bounds, rank, q_local, k_local, v_local, table_q, table_k, table_v, group=None):
    """The window-centric module (fused.cell_attention) for one scene over several ranks.  The unit of work is a CELL: rank r of w
    takes every w-th cell of the size-sorted list (CellPlan.share), so the shares are balanced without looking at the geometry.
    q/k/v_local are the rank's own rows (the same row bounds as sharded_window_attention); all three are all-gathered, the rank
    computes its cells' output rows, and a reduce-scatter hands every rank the sum's rows it owns.  Backward: grad_out
    all-gathered, grad_q / grad_k / grad_v reduce-scattered (through the gathers' backward), table gradients all-reduced.
    Returns out_local [hi-lo, h, 16]; every rank ends with the rows of the single-GPU result and gradient it owns."""
    world = len(bounds) - 1
    q_full = _GatherRows.apply(q_local, bounds, rank, group)
    k_full = _GatherRows.apply(k_local, bounds, rank, group)
    v_full = _GatherRows.apply(v_local, bounds, rank, group)
    tq, tk, tv = (_SharedParam.apply(t, group) for t in (table_q, table_k, table_v))
    part = cell_attention(q_full, k_full, v_full, tq, tk, tv, plan.share(rank, world) if world > 1 else plan)
    return _ScatterSumRows.apply(part, bounds, rank, group)


def sharded_window_attention(ops, shard, bounds, rank, q_local, k_local, v_local, table_q, table_k, table_v, n_max=0,
                             group=None, segment_softmax=None):
    """WindowAttention.forward's op sequence (model/stratified_transformer.py:183-208) for the queries of
    `shard`; q/k/v_local are the rank's own rows [hi-lo, h, d].  Returns out_local [hi-lo, h, d].
    Gradients: q/k/v_local receive exactly the rows of the single-GPU gradient; table gradients are
    all-reduced, i.e. every rank ends with the full table gradient."""
    k_full = _GatherRows.apply(k_local, bounds, rank, group)
    v_full = _GatherRows.apply(v_local, bounds, rank, group)
    tq, tk, tv = (_SharedParam.apply(t, group) for t in (table_q, table_k, table_v))
    softmax = segment_softmax or ops.segment_softmax
    a1 = ops.attention_step1_v2(q_local, k_full, shard.index_1, shard.offsets, n_max)
    a2 = ops.dot_prod_with_idx_v3(q_local, shard.offsets, n_max, k_full, shard.index_1, tq, tk, shard.rel_idx)
    sm = softmax(a1 + a2, shard.offsets)
    return ops.attention_step2_with_rel_pos_value_v2(sm, v_full, shard.offsets, n_max, shard.index_1, tv, shard.rel_idx)


# ---------------------------------------------------------------------------------------------------------------------
# Shard by WINDOW, exchange only boundary rows (north_star: "scenes shard by window ... all-gather of boundary keys";
# SURVEY 8e "sort by large-window id ... optionally only halo rows").
#
# Ownership: the points sorted by their (unshifted) large-window id; rank r owns a contiguous range of that order, cut so that
# the PAIR counts are balanced.  A rank's keys are then its own rows plus a HALO (the neighbouring windows' rows that its
# queries' shifted windows and stratified candidates reach), and only halo rows travel: one all_to_all_single per row tensor
# and direction (RCCL: every peer's slice goes straight to it over its own xGMI link), no padding, no concatenation.
# Everything stays in the GLOBAL row space: the rank's full-size buffers hold its own and its halo rows (the rest reads as zero and
# is never referenced by its pair list / its cells), so no index is renumbered and the kernels are the single-GPU ones.
# ---------------------------------------------------------------------------------------------------------------------
@dataclass
class HaloPlan:
    rank: int
    world: int
    n_points: int
    own_ids: torch.Tensor        # [n_own] i64: rows this rank owns (ownership order)
    need_ids: torch.Tensor       # [n_need] i64: foreign rows this rank touches, ordered by (owner, id)
    send_ids: torch.Tensor       # [n_send] i64: own rows other ranks touch, ordered by (touching rank, id)
    recv_splits: list            # rows received from every owner (sums to n_need)
    send_splits: list            # rows sent to every touching rank (sums to n_send)

    def halo_fraction(self):
        """rows that travel per exchanged tensor, as a fraction of what an all-gather of that tensor moves to this rank"""
        return float(sum(self.recv_splits)) / max(self.n_points - int(self.own_ids.shape[0]), 1)


def window_owners(large_partition, offsets, world):
    """(owner_of [N] i32, bounds [world+1], order): ownership by position in the large-window order, cut where the cumulative PAIR
    count crosses r * M / world.  offsets: the CSR offsets of a block pattern of the stage, or a list of them (plain and shifted
    pattern: their pair counts are added, so that the blocks of both patterns are balanced by the one ownership the stage has).
    large_partition: HipPartition / WindowPartition of the unshifted large windows (its `order` = points sorted by window)."""
    order = large_partition.order.long()
    offs_list = list(offsets) if isinstance(offsets, (list, tuple)) else [offsets]
    counts = sum((o[1:] - o[:-1]).to(torch.int64) for o in offs_list)
    cum = torch.cumsum(counts[order], 0)
    M = int(cum[-1])
    N = int(order.shape[0])
    targets = torch.arange(1, world, device=cum.device, dtype=torch.int64) * M // world
    cuts = torch.searchsorted(cum, targets, right=False).clamp_(0, N)
    bounds = [0] + [int(c) for c in cuts.tolist()] + [N]
    owner_sorted = torch.searchsorted(torch.tensor(bounds[1:], device=cum.device, dtype=torch.int64), torch.arange(N, device=cum.device), right=True)
    owner_of = torch.empty(N, dtype=torch.int32, device=cum.device)
    owner_of[order] = owner_sorted.to(torch.int32)
    return owner_of, bounds, order


def make_halo_plan(owner_of, touch_rank, touch_row, rank, world, own_ids):
    """touch_rank [T] / touch_row [T]: "rank r reads (or adds to) row j", for ALL ranks (the index is replicated, so every rank
    derives every rank's lists and the two ends of each transfer agree by construction).  One host sync (the split sizes)."""
    N = int(owner_of.shape[0])
    tr, row = touch_rank.long(), touch_row.long()
    own = owner_of.long()[row]
    foreign = tr != own
    key = torch.unique((tr[foreign] * world + own[foreign]) * N + row[foreign])          # sorted by (touching rank, owner, id)
    pair = key // N
    ids = key - pair * N
    counts = torch.bincount(pair, minlength=world * world).view(world, world).tolist()    # [touching rank][owner]
    t_rank, t_owner = pair // world, pair % world
    need_ids = ids[t_rank == rank]
    send_ids = ids[t_owner == rank]
    return HaloPlan(rank, world, N, own_ids, need_ids, send_ids, [counts[rank][o] for o in range(world)], [counts[t][rank] for t in range(world)])


def _all_to_all_rows(send, send_splits, recv_splits, group):
    tail = tuple(send.shape[1:])
    staged = send.is_cuda and _host_staged(group)
    src = send.cpu() if staged else send.contiguous()
    out = torch.empty((sum(recv_splits),) + tail, dtype=send.dtype, device=src.device)
    _count(src)
    _comm("all_to_all", dist.all_to_all_single, out, src, list(recv_splits), list(send_splits), group=group)
    return out.to(send.device)


class _ExchangeRows(torch.autograd.Function):
    """own rows [n_own, ...] -> [N, ...] with this rank's own and halo rows filled (the others zero);
    backward: the gradient's own rows plus what the other ranks accumulated for them (the reverse exchange)."""

    @staticmethod
    def forward(ctx, local, halo, group):
        ctx.halo, ctx.group = halo, group
        full = torch.zeros((halo.n_points,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        full[halo.own_ids] = local
        if halo.world > 1:
            full[halo.need_ids] = _all_to_all_rows(full[halo.send_ids], halo.send_splits, halo.recv_splits, group)
        return full

    @staticmethod
    def backward(ctx, grad_full):
        halo, group = ctx.halo, ctx.group
        g = grad_full[halo.own_ids].contiguous()
        if halo.world > 1:
            back = _all_to_all_rows(grad_full[halo.need_ids], halo.recv_splits, halo.send_splits, group)   # ordered like send_ids
            pos = torch.empty(halo.n_points, dtype=torch.int64, device=g.device)
            pos[halo.own_ids] = torch.arange(halo.own_ids.shape[0], device=g.device)
            g.index_add_(0, pos[halo.send_ids], back)
        return g, None, None


class _ReturnRows(torch.autograd.Function):
    """[N, ...] partial results (this rank's own rows and its contributions to foreign rows) -> the complete own rows;
    the mirror image of _ExchangeRows."""

    @staticmethod
    def forward(ctx, full, halo, group):
        ctx.halo, ctx.group = halo, group
        c = _Ctx()
        c.halo, c.group = halo, group
        return _ExchangeRows.backward(c, full)[0]

    @staticmethod
    def backward(ctx, g_local):
        return _ExchangeRows.forward(_Ctx(), g_local.contiguous(), ctx.halo, ctx.group), None, None


@dataclass
class HaloShard:
    halo: HaloPlan
    offsets: torch.Tensor      # [n_own+1] i32: the CSR rows of the own queries, in ownership order
    index_1: torch.Tensor      # [M_local] i32 GLOBAL key ids
    rel_idx: torch.Tensor      # [M_local, 3] i32


def make_halo_shard(block, owner_of, order, bounds, rank, world):
    """The operators' shard under window ownership: the own queries' rows of the block's CSR (gathered in ownership order) and
    the halo = foreign keys of those pairs."""
    own_ids = order[bounds[rank]:bounds[rank + 1]]
    offs = block.offsets.long()
    counts = offs[own_ids + 1] - offs[own_ids]
    local_offs = torch.zeros(own_ids.shape[0] + 1, dtype=torch.int64, device=offs.device)
    local_offs[1:] = torch.cumsum(counts, 0)
    pos = torch.repeat_interleave(offs[own_ids] - local_offs[:-1], counts) + torch.arange(int(local_offs[-1]), device=offs.device)
    halo = make_halo_plan(owner_of, owner_of[block.index_0.long()], block.index_1, rank, world, own_ids)
    return HaloShard(halo, local_offs.to(torch.int32), block.index_1[pos].contiguous(), block.rel_idx[pos].contiguous())


def halo_window_attention(ops, shard, q_local, k_local, v_local, table_q, table_k, table_v, n_max=0, group=None, segment_softmax=None):
    """sharded_window_attention with window ownership: q/k/v_local are the rank's own rows in ownership order; only the halo
    rows of k and v travel (forward) and only their gradients travel back."""
    k_full = _ExchangeRows.apply(k_local, shard.halo, group)
    v_full = _ExchangeRows.apply(v_local, shard.halo, group)
    tq, tk, tv = (_SharedParam.apply(t, group) for t in (table_q, table_k, table_v))
    softmax = segment_softmax or ops.segment_softmax
    a1 = ops.attention_step1_v2(q_local, k_full, shard.index_1, shard.offsets, n_max)
    a2 = ops.dot_prod_with_idx_v3(q_local, shard.offsets, n_max, k_full, shard.index_1, tq, tk, shard.rel_idx)
    sm = softmax(a1 + a2, shard.offsets)
    return ops.attention_step2_with_rel_pos_value_v2(sm, v_full, shard.offsets, n_max, shard.index_1, tv, shard.rel_idx)


def make_halo_cells(plan, owner_of, order, bounds, rank, world):
    """The window-centric kernels under window ownership: a cell belongs to the rank that owns its FIRST query; its other queries
    and its keys may be foreign rows (the halo: q / k / v and grad_out come in, out and the row gradients go back).
    -> (plan restricted to the rank's cells, HaloPlan)"""
    nC = plan.n_cells
    first_q = plan.cell_order.long()[plan.cell_qstart[:nC].long()]
    cell_owner = owner_of.long()[first_q]                                            # [nC]
    perm = plan.cell_perm[:nC].long()
    mine = perm[cell_owner[perm] == rank].to(torch.int32).contiguous()               # the rank's cells, largest tile first
    count = torch.tensor([mine.shape[0]], dtype=torch.int32, device=mine.device)
    # (a device tensor of the list's length would spare the host the sync of the boolean index; the list is built once per pattern)
    touch_rank = torch.cat([cell_owner[plan.qcell.long()], cell_owner[plan.kcell[:plan.n_keyslots].long()]])
    touch_row = torch.cat([plan.cell_order.long(), plan.cell_keys[:plan.n_keyslots].long()])
    own_ids = order[bounds[rank]:bounds[rank + 1]]
    halo = make_halo_plan(owner_of, touch_rank, touch_row, rank, world, own_ids)
    if mine.shape[0] == 0:
        mine = torch.zeros(1, dtype=torch.int32, device=count.device)
    return plan.with_tasks(mine, count), halo


def halo_cell_attention(cell_attention, plan_local, halo, q_local, k_local, v_local, table_q, table_k, table_v, group=None):
    """sharded_cell_attention with window ownership: the rank computes the cells it owns on full-size buffers that hold its own
    and its halo rows; out rows of foreign queries (boundary cells) return to their owners."""
    q_full = _ExchangeRows.apply(q_local, halo, group)
    k_full = _ExchangeRows.apply(k_local, halo, group)
    v_full = _ExchangeRows.apply(v_local, halo, group)
    tq, tk, tv = (_SharedParam.apply(t, group) for t in (table_q, table_k, table_v))
    part = cell_attention(q_full, k_full, v_full, tq, tk, tv, plan_local)
    return _ReturnRows.apply(part, halo, group)


def sharded_furthestsampling(fps, xyz, offset_host, new_offset_host, rank, world, group=None):
    """FPS of a BATCH over the ranks ("one rank per batch element then all-gather idx"): FPS is sequential inside an
    element but the elements are independent, so rank r samples elements r, r + world, ... and the index lists are gathered.
    fps(xyz, offset, new_offset) -> idx is the single-GPU operator; returns the same [new_offset[-1]] i32 tensor on every rank."""
    b = len(offset_host)
    starts = [0] + [int(o) for o in offset_host[:-1]]
    counts = [int(new_offset_host[i]) - (int(new_offset_host[i - 1]) if i else 0) for i in range(b)]
    mine = list(range(rank, b, world))
    dev = xyz.device
    if mine:
        sub = torch.cat([xyz[starts[i]:int(offset_host[i])] for i in mine]).contiguous()
        sizes = [int(offset_host[i]) - starts[i] for i in mine]
        sub_off = torch.tensor([sum(sizes[:j + 1]) for j in range(len(mine))], dtype=torch.int32, device=dev)
        sub_new = torch.tensor([sum(counts[i] for i in mine[:j + 1]) for j in range(len(mine))], dtype=torch.int32, device=dev)
        idx = fps(sub, sub_off, sub_new).long()
        shift = torch.repeat_interleave(torch.tensor([starts[i] - sum(sizes[:j]) for j, i in enumerate(mine)], device=dev),
                                        torch.tensor([counts[i] for i in mine], device=dev))
        idx = (idx + shift).to(torch.int32)
    else:
        idx = torch.zeros(0, dtype=torch.int32, device=dev)
    per_rank = [sum(counts[i] for i in range(r, b, world)) for r in range(world)]
    mx = max(per_rank + [1])
    staged = idx.is_cuda and _host_staged(group)
    pad = torch.zeros(mx, dtype=torch.int32, device="cpu" if staged else dev)
    pad[: idx.shape[0]] = idx
    gathered = torch.empty(world * mx, dtype=torch.int32, device=pad.device)
    _count(pad)
    _comm("all_gather", dist.all_gather_into_tensor, gathered, pad, group=group)
    gathered = gathered.view(world, mx).to(dev)
    out = torch.empty(int(new_offset_host[-1]), dtype=torch.int32, device=dev)
    cursor = [0] * world
    for i in range(b):
        r = i % world
        lo = int(new_offset_host[i]) - counts[i]
        out[lo:lo + counts[i]] = gathered[r, cursor[r]:cursor[r] + counts[i]]
        cursor[r] += counts[i]
    return out
