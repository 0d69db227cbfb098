"""Operator API of the Stratified Transformer hot path on MI355X.

Drop-in for the reference's `lib/pointops2/functions/pointops.py` (SURVEY.md §8b seam B1): the same
module-level callables, positional arguments, return shapes/dtypes and `backward` arities, written
from scratch over the C ABI of libpointops2_hip.so (through `pointops2_cuda`).  Citations are to
/root/reference/lib/pointops2/functions/pointops.py.

Differences, all compatible:
  * `n_max` may be an int or the 0-dim device tensor the model passes
    (model/stratified_transformer.py:315); it is never read on the host (no device sync) because
    the kernels do not size their workgroups by it and have no 1024-keys-per-query limit
    (the reference asserts `n_max <= 1024`, :150).
  * Backward passes do not scatter with global float atomics: a key-major (CSC) transposition of
    the pair list is built once per index pattern (cached, keyed by tensor identity) and the
    key-side gradients are gathered.
  * Inputs must be GPU tensors: there is no CPU fallback.
"""
import os
from collections import OrderedDict

import torch
from torch.autograd import Function

from . import _lib
from . import pointops2_cuda as pointops_cuda
from ._lib import ptr


def _zeros(shape, ref, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=ref.device)


def _nmax(n_max):
    return n_max if isinstance(n_max, int) else 0


def _check_pair_list(op, n_query_rows, offsets, index1, rel_idx=None, per_pair=None):
    """Shape facts every CSR operator relies on, checked on the host (no sync): the kernels walk `offsets[i] .. offsets[i + 1]`
    for i < n_query_rows and read index1 / rel_idx / the per-pair operand at those positions, so a pair list whose row count
    differs from the operand's rows makes them read past `offsets` (a garbage segment end = a wild read: the GPU memory fault of
    gpurun_out/r3_gpu_all2.log, where a test handed q rows of one cloud to the pair list of another).  The reference checks
    nothing here (pointops.py:446-482); stricter is compatible."""
    if offsets.dim() != 1 or int(offsets.shape[0]) - 1 != int(n_query_rows):
        raise ValueError(f"{op}: the pair list has {int(offsets.shape[0]) - 1} rows (index offsets [{int(offsets.shape[0])}]), the query-side operand {int(n_query_rows)}")
    M = int(index1.shape[0])
    if rel_idx is not None and (rel_idx.dim() != 2 or int(rel_idx.shape[0]) != M or int(rel_idx.shape[1]) != 3):
        raise ValueError(f"{op}: rel_idx must be [{M}, 3], got {tuple(rel_idx.shape)}")
    if per_pair is not None and int(per_pair.shape[0]) != M:
        raise ValueError(f"{op}: the per-pair operand has {int(per_pair.shape[0])} rows, the pair list {M} pairs")


# ---------------------------------------------------------------------------------------------
# key-major (CSC) view of a CSR pair list, shared by the backward kernels
# ---------------------------------------------------------------------------------------------
class _CSC:
    __slots__ = ("offsets", "pair", "query", "keep", "n_keys")

    def __init__(self, offsets, pair, query, keep):
        self.offsets, self.pair, self.query, self.keep = offsets, pair, query, keep


_CSC_CACHE = OrderedDict()
_CSC_CACHE_SIZE = 8
CSC_BUILDS = 0  # transpositions built so far (tests: one per block on the drop-in path)


def csc_of(index0_offsets, index1, n_keys, alias=None):
    """Returns the cached CSC of (index0_offsets, index1); builds it on the current stream if absent.

    The cache key is the identity (address + version) of the two index tensors; entries hold strong
    references to them, so an address cannot be recycled for different contents while cached.

    alias: the block's `n_max` tensor.  The unmodified model passes FRESH `index_1.int()` / `index_0_offsets.int()`
    copies to each of its three operators (model/stratified_transformer.py:183,194,208), so the identity of the index
    tensors never repeats inside a block - but the 0-dim `n_max` tensor (:315) is the same object in all three calls and a
    new one in every block: it identifies the block's pair list.  With it the three backward operators of a block share
    ONE transposition (and one cache entry) instead of building three.
    """
    M_, N_ = int(index1.shape[0]), int(index0_offsets.shape[0]) - 1
    akey = None
    if torch.is_tensor(alias):
        akey = ("n_max", alias.data_ptr(), alias._version, M_, N_, int(n_keys), index1.device.index)
        hit = _CSC_CACHE.get(akey)
        if hit is not None:
            _CSC_CACHE.move_to_end(akey)
            return hit
    key = (index0_offsets.data_ptr(), index0_offsets._version, index1.data_ptr(), index1._version,
           int(index1.shape[0]), int(n_keys), index1.device.index)
    hit = _CSC_CACHE.get(key)
    if hit is not None:
        _CSC_CACHE.move_to_end(key)
        if akey is not None:
            _CSC_CACHE[akey] = hit
        return hit
    M = int(index1.shape[0])
    N = int(index0_offsets.shape[0]) - 1
    dev = index1.device
    l = _lib.lib()
    offsets = torch.empty(n_keys + 1, dtype=torch.int32, device=dev)
    pair = torch.empty(M, dtype=torch.int32, device=dev)
    query = torch.empty(M, dtype=torch.int32, device=dev)
    if M > 0:
        nbytes = int(l.pointops2_csc_workspace_bytes(max(N, n_keys), M))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            l.pointops2_set_key_rows(int(n_keys))
            _lib.call("pointops2_csc_build", N, M, ptr(index0_offsets), ptr(index1), ptr(offsets), ptr(pair), ptr(query),
                      ptr(ws), nbytes, device=dev)
    else:
        offsets.zero_()
    csc = _CSC(offsets, pair, query, (index0_offsets, index1, alias))
    csc.n_keys = int(n_keys)
    _CSC_CACHE[key] = csc
    if akey is not None:
        _CSC_CACHE[akey] = csc
    global CSC_BUILDS
    CSC_BUILDS += 1
    while len(_CSC_CACHE) > _CSC_CACHE_SIZE:
        _CSC_CACHE.popitem(last=False)
    return csc


_LAST_CSR = {}


def remember_csr(offsets, M):
    """The CSR offsets the model's A1 / A2 call just used: its next call is scatter_softmax over the same pair list
    (model/stratified_transformer.py:183-205), which only gets the per-pair query ids - compat.scatter_softmax takes the
    offsets from here instead of rebuilding them (a host sync, a unique and a cumsum per block).

    Remembered as a HINT only: (tensor, its version, N, M).  The entry keeps the tensor alive (its address cannot be
    recycled), last_csr() drops it when it was written to since, and the shim verifies on the device that the offsets
    describe the index it was given (csr_matches) before any kernel walks segments with them."""
    _LAST_CSR[offsets.device.index] = (offsets, offsets._version, int(offsets.shape[0]) - 1, int(M))


def last_csr(device_index, M):
    """-> offsets [N+1] i32 remembered for this device if they belong to an M-pair list and were not modified since."""
    hit = _LAST_CSR.get(device_index)
    if hit is None:
        return None
    offsets, version, N, M_seen = hit
    if M_seen != int(M) or offsets._version != version or int(offsets.shape[0]) - 1 != N:
        return None
    return offsets


def csr_matches(offsets, index):
    """0-dim bool device tensor: do `offsets [N+1] i32` describe the per-pair query ids `index [M]` (i32 / i64)?  No host
    sync; the kernel touches nothing outside the two tensors whatever they hold (csrc/misc.hip csr_matches_kernel)."""
    _lib.check_tensor(offsets, torch.int32, "offsets")
    if index.dtype not in (torch.int32, torch.int64):
        raise TypeError(f"index: expected int32 or int64, got {index.dtype}")
    index = index.contiguous()
    if not index.is_cuda or index.device != offsets.device or offsets.dim() != 1 or index.dim() != 1 or offsets.shape[0] < 1:
        raise RuntimeError("csr_matches: offsets [N+1] and index [M] must be 1-d tensors on the same GPU")
    bad = torch.empty(1, dtype=torch.int32, device=offsets.device)
    with torch.cuda.device(offsets.device):
        _lib.call("pointops2_csr_matches_launcher", int(offsets.shape[0]) - 1, int(index.shape[0]), ptr(offsets), ptr(index),
                  1 if index.dtype == torch.int64 else 0, ptr(bad), device=offsets.device)
    return bad[0] == 0


_ROW_ORDER_CACHE = OrderedDict()
_ROW_ORDER_CACHE_SIZE = 16
ROW_ORDER = os.environ.get("P2_ROW_ORDER", "1") != "0"
ROW_ORDER_BUILDS = 0


def row_order_of(index0_offsets, index1, alias=None):
    """The rows of the pair list in window order (csrc/index.hip pointops2_row_order_launcher: sorted by their first partner), cached
    like the CSC by the identity (address + version) of the two index tensors - or of `alias`, the block's n_max tensor (csc_of);
    None when it is not worth having (small clouds) or switched off (P2_ROW_ORDER=0).  The operators' pair walkers take their rows
    in this order (csrc/common.h rows_in_order): same sums, neighbouring waves share their partners' rows in the second-level cache."""
    N = int(index0_offsets.shape[0]) - 1
    if not ROW_ORDER or N < 2048 or not index1.is_cuda:
        return None
    akey = None
    if torch.is_tensor(alias):
        akey = ("n_max", alias.data_ptr(), alias._version, int(index1.shape[0]), N, index1.device.index)
        hit = _ROW_ORDER_CACHE.get(akey)
        if hit is not None:
            _ROW_ORDER_CACHE.move_to_end(akey)
            return hit[0]
    key = (index0_offsets.data_ptr(), index0_offsets._version, index1.data_ptr(), index1._version, int(index1.shape[0]), N, index1.device.index)
    hit = _ROW_ORDER_CACHE.get(key)
    if hit is not None:
        _ROW_ORDER_CACHE.move_to_end(key)
        if akey is not None:
            _ROW_ORDER_CACHE[akey] = hit
        return hit[0]
    l = _lib.lib()
    order = torch.empty(N, dtype=torch.int32, device=index1.device)
    ws = torch.empty(int(l.pointops2_row_order_workspace_bytes(N)), dtype=torch.uint8, device=index1.device)
    _lib.call("pointops2_row_order_launcher", N, int(index1.shape[0]), ptr(index0_offsets), ptr(index1), ptr(order), ptr(ws), ws.numel(), device=index1.device)
    _ROW_ORDER_CACHE[key] = (order, index0_offsets, index1, alias)   # (the entry keeps the tensors alive: their addresses cannot be recycled)
    if akey is not None:
        _ROW_ORDER_CACHE[akey] = _ROW_ORDER_CACHE[key]
    global ROW_ORDER_BUILDS
    ROW_ORDER_BUILDS += 1
    while len(_ROW_ORDER_CACHE) > _ROW_ORDER_CACHE_SIZE:
        _ROW_ORDER_CACHE.popitem(last=False)
    return order


def seed_row_order(index0_offsets, index1, order, alias=None):
    """A caller that already has the rows of the pair list grouped by window (index_build.stage_index_hip: the small-window
    partition's point order IS such an order) hands it over; row_order_of then finds it instead of sorting."""
    N = int(index0_offsets.shape[0]) - 1
    if not ROW_ORDER or N < 2048 or order is None or int(order.shape[0]) != N:
        return
    entry = (order, index0_offsets, index1, alias)
    _ROW_ORDER_CACHE[(index0_offsets.data_ptr(), index0_offsets._version, index1.data_ptr(), index1._version, int(index1.shape[0]), N, index1.device.index)] = entry
    if torch.is_tensor(alias):
        _ROW_ORDER_CACHE[("n_max", alias.data_ptr(), alias._version, int(index1.shape[0]), N, index1.device.index)] = entry
    while len(_ROW_ORDER_CACHE) > _ROW_ORDER_CACHE_SIZE:
        _ROW_ORDER_CACHE.popitem(last=False)


class _with_rows:
    """the pair walkers of the launchers called inside take their rows in `order` (None: by index)"""

    def __init__(self, order):
        self.order = order

    def __enter__(self):
        if self.order is not None:
            _lib.lib().pointops2_set_row_order(ptr(self.order), int(self.order.shape[0]))

    def __exit__(self, *exc):
        if self.order is not None:
            _lib.lib().pointops2_set_row_order(None, 0)


def clear_caches():
    _LAST_CSR.clear()
    _CSC_CACHE.clear()
    _ROW_ORDER_CACHE.clear()
    _FPS_CACHE.clear()
    _HOST_OFFSETS.clear()


class _with_csc:
    def __init__(self, csc):
        self.csc = csc

    def __enter__(self):
        if self.csc is not None:
            l = _lib.lib()
            l.pointops2_set_csc(ptr(self.csc.offsets), ptr(self.csc.pair), ptr(self.csc.query))
            l.pointops2_set_key_rows(self.csc.n_keys)  # rows of k / v (may differ from the CSR's query rows)

    def __exit__(self, *exc):
        # always undone: a launcher called later through the plain reference API must not see a stale view
        if self.csc is not None:
            l = _lib.lib()
            l.pointops2_set_csc(None, None, None)
            l.pointops2_set_key_rows(0)


# ---------------------------------------------------------------------------------------------
# sampling / neighbours
# ---------------------------------------------------------------------------------------------
_FPS_CACHE = OrderedDict()
_FPS_CACHE_SIZE = 4
_HOST_OFFSETS = {}


def hint_host_offsets(tensor, values):
    """Tell the sampler the host copy of a (device) offset tensor, so that it need not read it back
    (a D2H copy synchronises the stream, which defeats running the sampler beside other work)."""
    _HOST_OFFSETS.pop((tensor.data_ptr(), tensor._version), None)  # (re-inserted at the young end)
    _HOST_OFFSETS[(tensor.data_ptr(), tensor._version)] = ([int(v) for v in values], tensor)
    while len(_HOST_OFFSETS) > 64:
        _HOST_OFFSETS.pop(next(iter(_HOST_OFFSETS)))


_UNORDERED = {}  # id(xyz) -> weakref: clouds their owner declared to be in no selection order (hint_unordered)


def hint_unordered(xyz):
    """Tell the sampler that `xyz` is a raw cloud, not the output of an earlier FPS kept in selection order: its calls on this
    tensor skip the identity-prefix probe (~60 us in front of the sampler).  Same samples with or without, also if the hint is wrong."""
    import weakref
    key = id(xyz)
    _UNORDERED[key] = weakref.ref(xyz, lambda _r, k=key: _UNORDERED.pop(k, None))


def _is_unordered(xyz):
    r = _UNORDERED.get(id(xyz))
    return r is not None and r() is xyz


def _host_list(t):
    hit = _HOST_OFFSETS.get((t.data_ptr(), t._version))
    return hit[0] if hit is not None else None


class FurthestSampling(Function):
    @staticmethod
    def forward(ctx, xyz, offset, new_offset):
        """:14-29  xyz (n,3) f32, offset (b) i32, new_offset (b) i32 -> idx (m) i32

        FPS is deterministic, so the n/8+1 samples BasicLayer asks for (model/stratified_transformer.py:289)
        are a prefix of the n/4+1 samples TransitionDown asks for on the same cloud (:103).  The sampler
        state of the last few clouds is kept (keyed by the identity of xyz, which the entry keeps alive):
        a repeated or shorter request is served from it, a longer one resumes it."""
        assert xyz.is_contiguous()
        n, b = xyz.shape[0], offset.shape[0]
        offs, new_offs = _host_list(offset), _host_list(new_offset)
        if offs is None or new_offs is None:
            offs, new_offs = torch.stack([offset, new_offset]).tolist()  # one D2H copy (the reference loops .item(), :22-25)
        n_max = offs[0]
        for i in range(1, b):
            n_max = max(offs[i] - offs[i - 1], n_max)
        want = [new_offs[0]] + [new_offs[i] - new_offs[i - 1] for i in range(1, b)]
        key = (xyz.data_ptr(), xyz._version, n, tuple(offs), xyz.device.index)
        ent = _FPS_CACHE.get(key)
        if ent is not None:
            _FPS_CACHE.move_to_end(key)
            have = ent["counts"]
            # the kept state may have been produced under another stream (a sampler prefetched beside the blocks): order behind it
            if ent.get("event") is not None and ent.get("stream") != torch.cuda.current_stream(xyz.device):
                torch.cuda.current_stream(xyz.device).wait_event(ent["event"])
            if all(w <= h for w, h in zip(want, have)):
                starts = [0] + ent["new_offs"][:-1]
                if b == 1:
                    return ent["idx"][: want[0]].clone()
                return torch.cat([ent["idx"][s: s + w] for s, w in zip(starts, want)])
            if not all(w >= h for w, h in zip(want, have)):
                ent = None  # neither a prefix nor a pure extension of the kept state: start over
        l = _lib.lib()
        idx = _zeros(new_offs[b - 1], xyz, torch.int32)
        tmp = torch.full((n,), 1e10, dtype=torch.float32, device=xyz.device)
        # lend the library scratch memory for the bucketed exact FPS (csrc/fps_bucket.hip); the
        # reference signature carries neither a workspace nor the total point count
        if ent is not None:
            ws = ent["ws"]
            l.pointops2_set_fps_resume(ptr(ent["idx"]), ptr(ent["new_offset"]))
        else:
            ws = torch.empty(int(l.pointops2_fps_workspace_bytes(b, n)), dtype=torch.uint8, device=xyz.device)
        l.pointops2_set_workspace(ptr(ws), ws.numel())
        l.pointops2_set_point_count(n)
        l.pointops2_set_fps_hint(1 if _is_unordered(xyz) else 0)
        try:
            pointops_cuda.furthestsampling_cuda(b, n_max, xyz, offset, new_offset, tmp, idx)
        finally:
            l.pointops2_set_workspace(None, 0)
            l.pointops2_set_fps_resume(None, None)
            l.pointops2_set_fps_hint(0)
        del tmp
        if n_max >= 2048:  # the bucketed kernel ran: its state can serve / resume later requests
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(xyz.device))
            _FPS_CACHE[key] = dict(xyz=xyz, ws=ws, idx=idx, new_offset=new_offset, new_offs=list(new_offs),
                                   counts=want, event=done, stream=torch.cuda.current_stream(xyz.device))
            while len(_FPS_CACHE) > _FPS_CACHE_SIZE:
                _FPS_CACHE.popitem(last=False)
            return idx.clone()
        return idx


furthestsampling = FurthestSampling.apply


def knn_squared(nsample, xyz, new_xyz, offset, new_offset):
    """exact kNN, ascending: (idx [m, nsample] i32, SQUARED distances [m, nsample] f32) - what the launcher returns (knnquery_cuda_kernel.cu:103-106)"""
    if new_xyz is None:
        new_xyz = xyz
    assert xyz.is_contiguous() and new_xyz.is_contiguous()
    m = new_xyz.shape[0]
    idx = _zeros((m, nsample), xyz, torch.int32)
    dist2 = _zeros((m, nsample), xyz)
    # lend scratch memory for the grid-accelerated exact search (csrc/knn_grid.hip)
    l = _lib.lib()
    n, b = xyz.shape[0], offset.shape[0]
    ws = torch.empty(int(l.pointops2_knn_workspace_bytes(n, m, b)), dtype=torch.uint8, device=xyz.device)
    l.pointops2_set_workspace(ptr(ws), ws.numel())
    l.pointops2_set_point_count(n)
    l.pointops2_set_batch_count(b)
    try:
        pointops_cuda.knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2)
    finally:
        l.pointops2_set_workspace(None, 0)
    return idx, dist2


class KNNQuery(Function):
    @staticmethod
    def forward(ctx, nsample, xyz, new_xyz, offset, new_offset):
        """:34-47  -> idx (m, nsample) i32, dist (m, nsample) f32 (Euclidean, sqrt applied here)"""
        idx, dist2 = knn_squared(nsample, xyz, new_xyz, offset, new_offset)
        return idx, torch.sqrt(dist2)


knnquery = KNNQuery.apply


def ball_query(radius, max_num, x, y, offset_x, offset_y):
    """The stem's neighbour search (train_backup.py:362-364: `tp.ball_query(radius, max_num, coord, coord, mode="partial_dense",
    batch_x=batch, batch_y=batch)[0]`, torch_points_kernels 0.6.10 - third-party, not under the reference: PARITY UNPINNED,
    restated from its published behaviour): for every point of y the up to `max_num` points of x of the same batch element
    within `radius` (squared distance < radius^2), nearest first, the row padded with -1.

    A ball query truncated to the nearest max_num IS the exact kNN(max_num) with the neighbours outside the ball masked,
    so it runs on the grid kNN kernels (csrc/knn_grid.hip).  x, y [n,3] / [m,3] f32 on the GPU, offsets i32 (the
    reference's cumulative batch ends instead of tp's per-point batch vectors).  Returns (idx [m, max_num] i32, dist2 f32)."""
    idx, d2 = knn_squared(max_num, x, y, offset_x, offset_y)
    r2 = torch.tensor(radius, dtype=torch.float32) * torch.tensor(radius, dtype=torch.float32)  # fp32 like the coordinates
    inside = d2 < r2.to(d2.device)
    return torch.where(inside, idx, torch.full_like(idx, -1)), torch.where(inside, d2, torch.full_like(d2, -1.0))


class Grouping(Function):
    @staticmethod
    def forward(ctx, input, idx):
        """:52-65  input (n,c), idx (m,nsample) -> (m,nsample,c)"""
        assert input.is_contiguous() and idx.is_contiguous()
        m, nsample, n, c = idx.shape[0], idx.shape[1], input.shape[0], input.shape[1]
        output = torch.empty((m, nsample, c), dtype=torch.float32, device=input.device)
        pointops_cuda.grouping_forward_cuda(m, nsample, c, input, idx, output)
        ctx.n = n
        ctx.save_for_backward(idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        n = ctx.n
        idx, = ctx.saved_tensors
        m, nsample, c = grad_output.shape
        grad_input = _zeros((n, c), grad_output)
        pointops_cuda.grouping_backward_cuda(m, nsample, c, grad_output.contiguous(), idx, grad_input)
        return grad_input, None


grouping = Grouping.apply


# ---------------------------------------------------------------------------------------------
# A1
# ---------------------------------------------------------------------------------------------
class AttentionStep1(Function):
    @staticmethod
    def forward(ctx, q, k, index0, index1):
        """:82-102  pair-indexed form -> (M, h)"""
        assert q.is_contiguous() and k.is_contiguous() and index0.is_contiguous() and index1.is_contiguous()
        N_q, h, C_div_h = q.shape
        N_k = k.shape[0]
        M = index0.shape[0]
        C = int(C_div_h * h)
        output = _zeros((M, h), q)
        pointops_cuda.attention_step1_forward_cuda(N_k, M, h, C, q, k, index0, index1, output)
        ctx.N_q, ctx.N_k, ctx.C = N_q, N_k, C
        ctx.save_for_backward(q, k, index0, index1)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        N_q, N_k, C = ctx.N_q, ctx.N_k, ctx.C
        q, k, index0, index1 = ctx.saved_tensors
        M, h = grad_output.shape
        grad_output = grad_output.contiguous()
        grad_q = _zeros((N_q, h, C // h), q)
        grad_k = _zeros((N_k, h, C // h), q)
        pointops_cuda.attention_step1_backward_cuda(N_q, M, h, C, grad_output, index0, index1, q, k, grad_q, grad_k)
        return grad_q, grad_k, None, None


attention_step1 = AttentionStep1.apply


class AttentionStep1_v2(Function):
    @staticmethod
    def forward(ctx, q, k, index1, index0_offsets, n_max):
        """:142-164  q,k (N,h,d) f32; index1 (M) i32; index0_offsets (N+1) i32 -> (M,h)"""
        assert q.is_contiguous() and k.is_contiguous() and index0_offsets.is_contiguous() and index1.is_contiguous()
        N_q, h, C_div_h = q.shape
        N_k = k.shape[0]
        M = index1.shape[0]
        C = int(C_div_h * h)
        _check_pair_list("attention_step1_v2", N_q, index0_offsets, index1)
        output = torch.empty((M, h), dtype=torch.float32, device=q.device)
        # the launcher's N is the number of CSR rows (queries); the reference passes N_k, which is the same
        # number in the model and would be wrong anywhere else (its kernel grid is one block per query)
        with _with_rows(row_order_of(index0_offsets, index1, n_max)):
            pointops_cuda.attention_step1_forward_cuda_v2(int(index0_offsets.shape[0]) - 1, M, h, C, _nmax(n_max), q, k, index0_offsets, index1, output)
        remember_csr(index0_offsets, M)
        ctx.N_q, ctx.N_k, ctx.C, ctx.n_max = N_q, N_k, C, n_max
        ctx.save_for_backward(q, k, index0_offsets, index1)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        """:166-201 -> grad_q, grad_k, None, None, None"""
        N_q, N_k, C = ctx.N_q, ctx.N_k, ctx.C
        q, k, index0_offsets, index1 = ctx.saved_tensors
        M, h = grad_output.shape
        grad_output = grad_output.contiguous()
        grad_q = torch.empty((N_q, h, C // h), dtype=torch.float32, device=q.device)
        grad_k = _zeros((N_k, h, C // h), q)
        with _with_csc(csc_of(index0_offsets, index1, N_k, ctx.n_max)), _with_rows(row_order_of(index0_offsets, index1, ctx.n_max)):
            pointops_cuda.attention_step1_backward_cuda_v2(int(index0_offsets.shape[0]) - 1, M, h, C, _nmax(ctx.n_max), grad_output, index0_offsets, index1, q, k, grad_q, grad_k)
        return grad_q, grad_k, None, None, None


attention_step1_v2 = AttentionStep1_v2.apply


# ---------------------------------------------------------------------------------------------
# plain AV (no rel-pos value)
# ---------------------------------------------------------------------------------------------
class AttentionStep2(Function):
    @staticmethod
    def forward(ctx, attn, v, index0, index1):
        """:207-228  -> (N_q, h, d) with N_q = index0.max()+1"""
        assert attn.is_contiguous() and v.is_contiguous() and index0.is_contiguous() and index1.is_contiguous()
        M, h = attn.shape
        N_q = index0.max().item() + 1
        N_v, h, C_div_h = v.shape
        C = int(C_div_h * h)
        output = _zeros((N_q, h, C // h), v)
        pointops_cuda.attention_step2_forward_cuda(N_q, M, h, C, attn, v, index0, index1, output)
        ctx.M = M
        ctx.save_for_backward(attn, v, index0, index1)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        M = ctx.M
        attn, v, index0, index1 = ctx.saved_tensors
        N_v = v.shape[0]
        N_q, h, C_div_h = grad_output.shape
        C = h * C_div_h
        grad_output = grad_output.contiguous()
        grad_attn = _zeros((M, h), v)
        grad_v = _zeros((N_v, h, C // h), v)
        pointops_cuda.attention_step2_backward_cuda(N_q, M, h, C, grad_output, index0, index1, attn, v, grad_attn, grad_v)
        return grad_attn, grad_v, None, None


attention_step2 = AttentionStep2.apply


class AttentionStep2_v2(Function):
    """:268-316 — identical math; the reference routes it to the v1 kernel (:284)."""

    @staticmethod
    def forward(ctx, attn, v, index0, index1):
        return AttentionStep2.forward(ctx, attn, v, index0, index1)

    @staticmethod
    def backward(ctx, grad_output):
        return AttentionStep2.backward(ctx, grad_output)


attention_step2_v2 = AttentionStep2_v2.apply


# ---------------------------------------------------------------------------------------------
# A2
# ---------------------------------------------------------------------------------------------
class DotProdWithIdx(Function):
    @staticmethod
    def forward(ctx, q, index, table, rel_idx):
        """:320-335  single-table pair-indexed form -> (M, h)"""
        assert q.is_contiguous() and index.is_contiguous() and table.is_contiguous() and rel_idx.is_contiguous()
        N, h, hdim = q.shape
        M = index.shape[0]
        output = _zeros((M, h), q)
        pointops_cuda.dot_prod_with_idx_forward_cuda(N, M, h, hdim, q, index, table, rel_idx, output)
        ctx.save_for_backward(q, index, table, rel_idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        q, index, table, rel_idx = ctx.saved_tensors
        M, h = grad_output.shape
        N, _, hdim = q.shape
        L = table.shape[0]
        grad_output = grad_output.contiguous()
        grad_q = _zeros((N, h, hdim), q)
        grad_table = _zeros((L, h, hdim, 3), q)
        pointops_cuda.dot_prod_with_idx_backward_cuda(N, M, h, hdim, grad_output, q, index, table, rel_idx, grad_q, grad_table)
        return grad_q, None, grad_table, None


dot_prod_with_idx = DotProdWithIdx.apply


class DotProdWithIdx_v2(Function):
    @staticmethod
    def forward(ctx, q, index_q, k, index_k, table_q, table_k, rel_idx):
        """:372-406  bucketed form.  The host-side sort by merged rel index (:387-393) only drives the
        reference's work distribution; the result does not depend on it, so it is not computed here."""
        assert q.is_contiguous() and index_q.is_contiguous() and k.is_contiguous() and index_k.is_contiguous() \
            and table_q.is_contiguous() and table_k.is_contiguous() and rel_idx.is_contiguous()
        N, h, hdim = q.shape
        M = index_q.shape[0]
        L = table_q.shape[0]
        assert table_k.shape[0] == L and index_k.shape[0] == M
        output = _zeros((M, h), q)
        pointops_cuda.dot_prod_with_idx_forward_cuda_v2(N, M, h, hdim, 0, 0, q, index_q, k, index_k, table_q, table_k, rel_idx, None, None, output)
        ctx.save_for_backward(q, index_q, k, index_k, table_q, table_k, rel_idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        q, index_q, k, index_k, table_q, table_k, rel_idx = ctx.saved_tensors
        M, h = grad_output.shape
        N, _, hdim = q.shape
        L = table_q.shape[0]
        grad_output = grad_output.contiguous()
        grad_q, grad_k = _zeros((N, h, hdim), q), _zeros((N, h, hdim), q)
        grad_table_q, grad_table_k = _zeros((L, h, hdim, 3), q), _zeros((L, h, hdim, 3), q)
        pointops_cuda.dot_prod_with_idx_backward_cuda_v2(N, M, h, hdim, 0, 0, grad_output, q, index_q, k, index_k, table_q, table_k, rel_idx,
                                                         None, None, grad_q, grad_k, grad_table_q, grad_table_k)
        return grad_q, None, grad_k, None, grad_table_q, grad_table_k, None


dot_prod_with_idx_v2 = DotProdWithIdx_v2.apply


class DotProdWithIdx_v3(Function):
    @staticmethod
    def forward(ctx, q, index_q_offsets, n_max, k, index_k, table_q, table_k, rel_idx):
        """:446-482  -> (M, h)"""
        assert q.is_contiguous() and index_q_offsets.is_contiguous() and k.is_contiguous() and index_k.is_contiguous() \
            and table_q.is_contiguous() and table_k.is_contiguous() and rel_idx.is_contiguous()
        N, h, hdim = q.shape
        M = index_k.shape[0]
        L = table_q.shape[0]
        assert table_k.shape[0] == L
        _check_pair_list("dot_prod_with_idx_v3", N, index_q_offsets, index_k, rel_idx)
        output = torch.empty((M, h), dtype=torch.float32, device=q.device)
        with _with_rows(row_order_of(index_q_offsets, index_k, n_max)):
            pointops_cuda.dot_prod_with_idx_forward_cuda_v3(N, M, h, hdim, _nmax(n_max), q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output)
        remember_csr(index_q_offsets, M)
        ctx.n_max = n_max
        ctx.save_for_backward(q, index_q_offsets, k, index_k, table_q, table_k, rel_idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        """:484-517 -> grad_q, None, None, grad_k, None, grad_table_q, grad_table_k, None"""
        q, index_q_offsets, k, index_k, table_q, table_k, rel_idx = ctx.saved_tensors
        M, h = grad_output.shape
        N, _, hdim = q.shape
        L = table_q.shape[0]
        grad_output = grad_output.contiguous()
        grad_q = torch.empty((N, h, hdim), dtype=torch.float32, device=q.device)
        grad_k = _zeros((k.shape[0], h, hdim), q)
        grad_table_q, grad_table_k = _zeros((L, h, hdim, 3), q), _zeros((L, h, hdim, 3), q)
        with _with_csc(csc_of(index_q_offsets, index_k, k.shape[0], ctx.n_max)), _with_rows(row_order_of(index_q_offsets, index_k, ctx.n_max)):
            pointops_cuda.dot_prod_with_idx_backward_cuda_v3(N, M, h, hdim, _nmax(ctx.n_max), grad_output, q, index_q_offsets, k, index_k,
                                                             table_q, table_k, rel_idx, grad_q, grad_k, grad_table_q, grad_table_k)
        return grad_q, None, None, grad_k, None, grad_table_q, grad_table_k, None


dot_prod_with_idx_v3 = DotProdWithIdx_v3.apply


# ---------------------------------------------------------------------------------------------
# A4
# ---------------------------------------------------------------------------------------------
class AttentionStep2WithRelPosValue(Function):
    @staticmethod
    def forward(ctx, attn, v, index0, index1, table, rel_idx):
        """:521-540  pair-indexed form -> (N_q, h, hdim)"""
        assert attn.is_contiguous() and v.is_contiguous() and index0.is_contiguous() and index1.is_contiguous() \
            and table.is_contiguous() and rel_idx.is_contiguous()
        M, h = attn.shape
        N_v, h, hdim = v.shape
        N_q = index0.max().item() + 1
        output = _zeros((N_q, h, hdim), v)
        pointops_cuda.attention_step2_with_rel_pos_value_forward_cuda(N_q, M, h, hdim, attn, v, index0, index1, table, rel_idx, output)
        ctx.save_for_backward(attn, v, index0, index1, table, rel_idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        attn, v, index0, index1, table, rel_idx = ctx.saved_tensors
        N_q, h, hdim = grad_output.shape
        N_v, M, L = v.shape[0], attn.shape[0], table.shape[0]
        grad_output = grad_output.contiguous()
        grad_attn, grad_v, grad_table = _zeros((M, h), v), _zeros((N_v, h, hdim), v), _zeros((L, h, hdim, 3), v)
        pointops_cuda.attention_step2_with_rel_pos_value_backward_cuda(N_q, M, h, hdim, grad_output, index0, index1, attn, v, table, rel_idx,
                                                                       grad_attn, grad_v, grad_table)
        return grad_attn, grad_v, None, None, grad_table, None


attention_step2_with_rel_pos_value = AttentionStep2WithRelPosValue.apply


class AttentionStep2WithRelPosValue_v2(Function):
    @staticmethod
    def forward(ctx, attn, v, index0_offsets, n_max, index1, table, rel_idx):
        """:584-604  attn (M,h), v (N,h,hdim) -> (N,h,hdim)"""
        assert attn.is_contiguous() and v.is_contiguous() and index0_offsets.is_contiguous() and index1.is_contiguous() \
            and table.is_contiguous() and rel_idx.is_contiguous()
        M, h = attn.shape
        N_v, h, hdim = v.shape
        N = int(index0_offsets.shape[0]) - 1  # CSR rows = queries (== N_v in the model, :594-597)
        _check_pair_list("attention_step2_with_rel_pos_value_v2", N, index0_offsets, index1, rel_idx, attn)
        output = torch.empty((N, h, hdim), dtype=torch.float32, device=v.device)
        with _with_rows(row_order_of(index0_offsets, index1, n_max)):
            pointops_cuda.attention_step2_with_rel_pos_value_forward_cuda_v2(N, M, h, hdim, _nmax(n_max), attn, v, index0_offsets, index1, table, rel_idx, output)
        ctx.n_max = n_max
        ctx.save_for_backward(attn, v, index0_offsets, index1, table, rel_idx)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        """:606-644 -> grad_attn, grad_v, None, None, None, grad_table, None (grad_output must be contiguous, :621)"""
        attn, v, index0_offsets, index1, table, rel_idx = ctx.saved_tensors
        N_v, h, hdim = v.shape
        N = int(index0_offsets.shape[0]) - 1
        M, L = attn.shape[0], table.shape[0]
        assert grad_output.is_contiguous()
        grad_attn = torch.empty((M, h), dtype=torch.float32, device=v.device)
        grad_v, grad_table = _zeros((N_v, h, hdim), v), _zeros((L, h, hdim, 3), v)
        with _with_csc(csc_of(index0_offsets, index1, N_v, ctx.n_max)), _with_rows(row_order_of(index0_offsets, index1, ctx.n_max)):
            pointops_cuda.attention_step2_with_rel_pos_value_backward_cuda_v2(N, M, h, hdim, _nmax(ctx.n_max), grad_output, index0_offsets, index1,
                                                                              attn, v, table, rel_idx, grad_attn, grad_v, grad_table)
        return grad_attn, grad_v, None, None, None, grad_table, None


attention_step2_with_rel_pos_value_v2 = AttentionStep2WithRelPosValue_v2.apply


# ---------------------------------------------------------------------------------------------
# A3 (not part of the reference module: the model takes it from torch_scatter)
# ---------------------------------------------------------------------------------------------
class SegmentSoftmax(Function):
    """softmax over each CSR segment of src (M, h), per head."""

    @staticmethod
    def forward(ctx, src, offsets, valid=None):
        src = src.contiguous()
        _lib.check_tensor(src, torch.float32, "src")
        _lib.check_tensor(offsets, torch.int32, "offsets")
        if src.dim() != 2 or offsets.dim() != 1 or offsets.shape[0] < 1:
            raise RuntimeError("segment_softmax: src must be [M, h], offsets [N+1]")
        M, h = src.shape
        N = offsets.shape[0] - 1
        out = torch.empty_like(src)
        with torch.cuda.device(src.device):
            _lib.call("segment_softmax_forward_launcher", N, M, h, ptr(src), ptr(offsets), ptr(out), device=src.device)
        if valid is not None and M > 0:  # a device-side verdict (compat.scatter_softmax): poison instead of a host round trip
            out[0] = torch.where(valid, out[0], torch.full_like(out[0], float("nan")))
        ctx.save_for_backward(out, offsets)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        out, offsets = ctx.saved_tensors
        M, h = out.shape
        N = offsets.shape[0] - 1
        grad_out = grad_out.contiguous()
        _lib.check_tensor(grad_out, torch.float32, "grad_out")
        grad_src = torch.empty_like(out)
        with torch.cuda.device(out.device):
            _lib.call("segment_softmax_backward_launcher", N, M, h, ptr(out), ptr(grad_out), ptr(offsets), ptr(grad_src), device=out.device)
        return grad_src, None, None


segment_softmax = SegmentSoftmax.apply


# ---------------------------------------------------------------------------------------------
# helpers around kNN (:648-693, :756-833)
# ---------------------------------------------------------------------------------------------
def queryandgroup(nsample, xyz, new_xyz, feat, idx, offset, new_offset, use_xyz=True, return_indx=False):
    """:648-675  -> (m, nsample, 3+c) or (m, nsample, c)"""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    if new_xyz is None:
        new_xyz = xyz
    if idx is None:
        idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    n, m, c = xyz.shape[0], new_xyz.shape[0], feat.shape[1]
    flat = idx.view(-1).long()
    grouped_feat = feat[flat, :].view(m, nsample, c)
    if use_xyz:
        grouped_xyz = xyz[flat, :].view(m, nsample, 3)
        grouped_xyz -= new_xyz.unsqueeze(1)
        out = torch.cat((grouped_xyz, grouped_feat), -1)
    else:
        out = grouped_feat
    return (out, idx) if return_indx else out


def Divide2Patch(nsample, xyz, offset, return_offset=False, anchor_scale=None):
    """:678-693"""
    downsample_scale = anchor_scale or nsample
    offs = offset.tolist()
    new_offset, count = [offs[0] // downsample_scale], offs[0] // downsample_scale
    for i in range(1, len(offs)):
        count += (offs[i] - offs[i - 1]) // downsample_scale
        new_offset.append(count)
    new_offset = torch.tensor(new_offset, dtype=torch.int32, device=xyz.device)
    idx = furthestsampling(xyz, offset, new_offset)
    new_xyz = xyz[idx.long()]
    p_idx, _ = knnquery(nsample, xyz, new_xyz, offset, new_offset)
    return (p_idx, new_offset) if return_offset else p_idx


class _WeightedGather(Function):
    """out[n, :] = sum_i weight[n, i] * input[idx[n, i], :], accumulated in the order i = 0 .. k-1 from zero (what the reference's torch
    loop :767-769 computes, bit for bit) - by the interpolation kernels (interpolation_cuda_kernel.cu:5-33) instead of k gathers,
    k multiplies and k adds over [n, c] temporaries."""

    @staticmethod
    def forward(ctx, input, idx, weight):
        input, idx, weight = input.contiguous(), idx.contiguous(), weight.contiguous()
        n, k = idx.shape
        m, c = input.shape
        output = _zeros((n, c), input)
        pointops_cuda.interpolation_forward_cuda(n, c, k, input, idx, weight, output)
        ctx.m = m
        ctx.save_for_backward(input, idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        input, idx, weight = ctx.saved_tensors
        n, k = idx.shape
        grad_output = grad_output.contiguous()
        grad_input = grad_weight = None
        if ctx.needs_input_grad[0]:
            grad_input = _zeros((ctx.m, grad_output.shape[1]), grad_output)
            pointops_cuda.interpolation_backward_cuda(n, grad_output.shape[1], k, grad_output, idx, weight, grad_input)
        if ctx.needs_input_grad[2]:  # (only interpolation_v2 with coordinates that require a gradient)
            grad_weight = torch.stack([(grad_output * input[idx[:, i].long(), :]).sum(-1) for i in range(k)], 1)
        return grad_input, None, grad_weight


def _inverse_distance_weights(dist):
    dist_recip = 1.0 / (dist + 1e-8)
    return dist_recip / torch.sum(dist_recip, dim=1, keepdim=True)


def interpolation(xyz, new_xyz, feat, offset, new_offset, k=3):
    """:756-770  inverse-distance weighted k-NN interpolation of feat (m, c) onto new_xyz (n, 3) -> (n, c); differentiable w.r.t. feat"""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
    return _WeightedGather.apply(feat, idx, _inverse_distance_weights(dist))


def interpolation_v2(xyz, new_xyz, feat, offset, new_offset, k=3):
    """:773-797  the same with the distances recomputed by torch (differentiable w.r.t. the coordinates)"""
    assert xyz.is_contiguous() and new_xyz.is_contiguous() and feat.is_contiguous()
    idx, _ = knnquery(k, xyz, new_xyz, offset, new_offset)
    dist = torch.sqrt(((new_xyz.unsqueeze(1) - xyz[idx.long()]) ** 2).sum(-1) + 1e-8)
    return _WeightedGather.apply(feat, idx, _inverse_distance_weights(dist))


class Interpolation(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, input, offset, new_offset, k=3):
        """:800-818"""
        assert xyz.is_contiguous() and new_xyz.is_contiguous() and input.is_contiguous()
        idx, dist = knnquery(k, xyz, new_xyz, offset, new_offset)
        dist_recip = 1.0 / (dist + 1e-8)
        norm = torch.sum(dist_recip, dim=1, keepdim=True)
        weight = (dist_recip / norm).contiguous()
        n, c, m = new_xyz.shape[0], input.shape[1], input.shape[0]
        output = _zeros((n, c), input)
        pointops_cuda.interpolation_forward_cuda(n, c, k, input, idx, weight, output)
        ctx.m, ctx.k = m, k
        ctx.save_for_backward(idx, weight)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        m, k = ctx.m, ctx.k
        idx, weight = ctx.saved_tensors
        n, c = grad_output.shape
        grad_input = _zeros((m, c), grad_output)
        pointops_cuda.interpolation_backward_cuda(n, c, k, grad_output.contiguous(), idx, weight, grad_input)
        return None, None, grad_input, None, None, None


interpolation2 = Interpolation.apply


class Subtraction(Function):
    @staticmethod
    def forward(ctx, input1, input2, idx):
        raise NotImplementedError("subtraction: Point-Transformer op outside the Stratified hot path (SURVEY.md §8)")


class Aggregation(Function):
    @staticmethod
    def forward(ctx, input, position, weight, idx):
        raise NotImplementedError("aggregation: Point-Transformer op outside the Stratified hot path (SURVEY.md §8)")


subtraction = Subtraction.apply
aggregation = Aggregation.apply
