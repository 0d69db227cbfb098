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
