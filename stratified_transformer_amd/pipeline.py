"""One pass of the StratifiedAttention hot path over a scene — the unit bench.py times
 and the full-size tests exercise.

For each stage s of a Stratified config (window w_s, quant q_s, C_s, h_s, depth_s):
    index build   four window partitions (grid_sample x4), stratified FPS (n//scale+1 per batch
                  element), pair list + rel-pos index for the even and the odd block pattern
    attention     depth_s x [A1, A2, add, A3, A4] forward and backward through the operator API
    transition    TransitionDown's FPS (ratio) + kNN(k) producing stage s+1's points
    upsample      kNN(k=3) of Upsample between stage s+1 and s
Linear layers (qkv/proj/MLP) are not part of the unit.
"""
from dataclasses import dataclass, field

import os
import torch

from . import index_build
from . import pointops as P


@dataclass
class StageConfig:
    window_size: float
    quant_size: float
    channels: int
    num_heads: int
    depth: int


@dataclass
class SceneConfig:
    name: str
    stages: list
    downsample_scale: int = 8
    ratio: float = 0.25
    k: int = 16
    up_k: int = 3
    stem_transformer: bool = True


def s3dis_config():
    """config/s3dis/s3dis_stratified_transformer.yaml:14-35 with the derivations of train.py:110-113"""
    grid, win, quant = 0.04, 4, 0.01
    ch, heads, depths = [48, 96, 192, 384], [3, 6, 12, 24], [2, 2, 6, 2]
    return SceneConfig("s3dis_stratified_transformer",
                       [StageConfig(grid * win * 2 ** i, quant * 2 ** i, ch[i], heads[i], depths[i]) for i in range(4)],
                       downsample_scale=8, ratio=0.25, k=16, up_k=3, stem_transformer=True)


def scannet_config():
    """config/scannetv2/scannetv2_stratified_transformer.yaml:13-34; stem_transformer False => attention
    starts at stage 1 after a TransitionDown (model/stratified_transformer.py:411-417)"""
    grid, win, quant = 0.02, 5, 0.005
    ch, heads, depths = [48, 96, 192, 384, 384], [3, 6, 12, 24, 24], [3, 3, 9, 3, 3]
    return SceneConfig("scannetv2_stratified_transformer",
                       [StageConfig(grid * win * 2 ** i, quant * 2 ** i, ch[i], heads[i], depths[i]) for i in range(5)],
                       downsample_scale=4, ratio=0.25, k=16, up_k=3, stem_transformer=False)


@dataclass
class StageState:
    """Resident tensors of one stage (synthetic q/k/v/tables/grad_out stand in for the Linear layers)."""
    xyz: torch.Tensor
    offset: torch.Tensor
    q: torch.Tensor = None
    k: torch.Tensor = None
    v: torch.Tensor = None
    tables: list = field(default_factory=list)
    grad_out: torch.Tensor = None
    window_quant: tuple = None  # (window_size, quant_size) of the stage (model_call_order_block evaluates :186-188 with them)
    bf16: list = None  # q / k / v / tables stored as bf16 (attention_block(fused="cell_bf16")), made on first use


class Timer:
    """Per-component device timing with events on torch's current stream (the stream every HIP launch
    of this package uses).  Disabled timers cost nothing."""

    def __init__(self, enabled=True, only=None):
        self.enabled = enabled
        self.only = only  # optional tuple of name prefixes: everything else runs untimed
        self.spans = []
        self.stage = None  # scene_pass notes the stage whose blocks it is enqueueing (per-stage totals)

    def run(self, name, fn, *a, **k):
        if not self.enabled or (self.only is not None and not name.startswith(self.only)):
            return fn(*a, **k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        self.spans.append((name, e0, e1, self.stage))
        return out

    def totals(self):
        """-> {name: (total_ms, calls)}; call after torch.cuda.synchronize()"""
        out = {}
        for name, e0, e1, _ in self.spans:
            t, c = out.get(name, (0.0, 0))
            out[name] = (t + e0.elapsed_time(e1), c + 1)
        return out

    def totals_by_stage(self, prefix):
        """-> {stage: (total_ms, calls)} of the spans whose name starts with `prefix` and that were enqueued under a noted stage"""
        out = {}
        for name, e0, e1, stage in self.spans:
            if stage is not None and name.startswith(prefix):
                t, c = out.get(stage, (0.0, 0))
                out[stage] = (t + e0.elapsed_time(e1), c + 1)
        return out


def table_rows(st):
    return 2 * int((2 * st.window_size + 1e-4) // st.quant_size)  # model/stratified_transformer.py:142,145


def make_stage_state(xyz, offset, st, seed):
    g = torch.Generator(device=xyz.device).manual_seed(seed)
    n, h, d = xyz.shape[0], st.num_heads, st.channels // st.num_heads
    L = table_rows(st)

    def rn(*shape, scale=1.0):
        return (torch.randn(*shape, generator=g, device=xyz.device) * scale).requires_grad_(True)

    return StageState(xyz, offset, rn(n, h, d), rn(n, h, d), rn(n, h, d),
                      [rn(L, h, d, 3, scale=0.02) for _ in range(3)],
                      torch.randn(n, h, d, generator=g, device=xyz.device), (st.window_size, st.quant_size))


def attention_block(state, blk, timer, fused=False, shard=None):
    """WindowAttention.forward's op sequence (model/stratified_transformer.py:183-208) + its backward.
    fused=True: the optional one-function fast path (fused.window_attention) instead of the five
    operators; fused="cell": the window-centric kernels (fused.cell_attention); same numbers.
    shard=(rank, world): this rank's query range of the block only (sharding.py): k / v rows all-gathered,
    their gradients reduce-scattered, table gradients all-reduced; returns the rank's rows of the output."""
    q, k, v = state.q, state.k, state.v
    tq, tk, tv = state.tables
    for t in (q, k, v, tq, tk, tv):
        t.grad = None
    if shard is not None and len(shard) > 2 and shard[2] == "halo":
        # ownership by window, only boundary rows travel (sharding.py, second half); the rank's rows are gathered from / scattered
        # into the resident full-size synthetic tensors here - in a model they would simply live in ownership order
        from . import sharding
        rank, world = shard[0], shard[1]
        if blk.owner is None:  # (scene_pass sets one ownership per stage, balanced over both patterns; a block on its own: its own pairs)
            blk.owner = sharding.window_owners(blk.parts["large"], blk.offsets, world)
        owner_of, bounds, order = blk.owner
        kind = "cell" if fused == "cell" else "ops"
        if blk.halo is None or blk.halo[1] != (rank, world, kind):
            made = sharding.make_halo_cells(blk.cells, owner_of, order, bounds, rank, world) if kind == "cell" \
                else sharding.make_halo_shard(blk, owner_of, order, bounds, rank, world)
            blk.halo = (made, (rank, world, kind))
        made = blk.halo[0]
        own = (made[1] if kind == "cell" else made.halo).own_ids
        ql, kl, vl = (t.detach()[own].requires_grad_(True) for t in (q, k, v))
        sharding.set_timer(timer)
        try:
            if kind == "cell":
                from . import fused as F
                out = timer.run("attn_fwd/halo_cell", sharding.halo_cell_attention, F.cell_attention, made[0], made[1], ql, kl, vl, tq, tk, tv)
            else:
                out = timer.run("attn_fwd/halo", sharding.halo_window_attention, P, made, ql, kl, vl, tq, tk, tv, blk.n_max)
            timer.run("attn_bwd", out.backward, state.grad_out[own])
        finally:
            sharding.set_timer(None)
        for full, loc in ((q, ql), (k, kl), (v, vl)):   # the rank's gradient rows, where the single-GPU pass has them
            full.grad = torch.zeros_like(full)
            full.grad[own] = loc.grad
        blk.halo_rows = own
        return out
    if shard is not None:
        from . import sharding
        rank, world = shard[0], shard[1]
        if getattr(blk, "shard", None) is None or blk.shard[2] != (rank, world):
            sh, bounds = sharding.make_shard(blk, rank, world)
            blk.shard = (sh, bounds, (rank, world))
        sh, bounds, _ = blk.shard
        sharding.set_timer(timer)
        try:
            if fused == "cell":  # the window-centric kernels on this rank's share of the cells (same row bounds)
                from . import fused as F
                out = timer.run("attn_fwd/sharded_cell", sharding.sharded_cell_attention, F.cell_attention, blk.cells, bounds, rank,
                                q[sh.lo:sh.hi], k[sh.lo:sh.hi], v[sh.lo:sh.hi], tq, tk, tv)
            else:
                out = timer.run("attn_fwd/sharded", sharding.sharded_window_attention, P, sh, bounds, rank, q[sh.lo:sh.hi], k[sh.lo:sh.hi], v[sh.lo:sh.hi],
                                tq, tk, tv, blk.n_max)
            timer.run("attn_bwd", out.backward, state.grad_out[sh.lo:sh.hi])
        finally:
            sharding.set_timer(None)
        return out
    if fused == "cell":
        from . import fused as F
        out = timer.run("attn_fwd/cell", F.cell_attention, q, k, v, tq, tk, tv, blk.cells)
        timer.run("attn_bwd", out.backward, state.grad_out)
        return out
    if fused == "cell_fwd":  # BASELINE config 2: forward only
        from . import fused as F
        with torch.no_grad():
            return timer.run("attn_fwd/cell", F.cell_attention, q, k, v, tq, tk, tv, blk.cells)
    if fused == "cell_bf16":  # BASELINE config 3, second leg: q / k / v / tables STORED as bf16 (fp32 arithmetic, fp32 out and sums)
        from . import fused as F
        if getattr(state, "bf16", None) is None:
            state.bf16 = [t.detach().to(torch.bfloat16).requires_grad_(True) for t in (q, k, v, tq, tk, tv)]
        for t in state.bf16:
            t.grad = None
        out = timer.run("attn_fwd/cell", F.cell_attention, *state.bf16, blk.cells)
        timer.run("attn_bwd", out.backward, state.grad_out)
        return out
    if fused == "model":
        return model_call_order_block(state, blk, timer)
    if fused:
        from . import fused as F
        out = timer.run("attn_fwd/fused", F.window_attention, q, k, v, tq, tk, tv, blk.offsets, blk.index_1, blk.rel_idx)
        timer.run("attn_bwd", out.backward, state.grad_out)
        return out
    a1 = timer.run("attn_fwd/A1", P.attention_step1_v2, q, k, blk.index_1, blk.offsets, blk.n_max)
    a2 = timer.run("attn_fwd/A2", P.dot_prod_with_idx_v3, q, blk.offsets, blk.n_max, k, blk.index_1, tq, tk, blk.rel_idx)
    s = timer.run("attn_fwd/add", torch.add, a1, a2)
    sm = timer.run("attn_fwd/A3", P.segment_softmax, s, blk.offsets)
    out = timer.run("attn_fwd/A4", P.attention_step2_with_rel_pos_value_v2, sm, v, blk.offsets, blk.n_max, blk.index_1, tv, blk.rel_idx)
    timer.run("attn_bwd", out.backward, state.grad_out)
    return out


def model_call_order_block(state, blk, timer, window_size=None, quant_size=None):
    """One block the way the UNMODIFIED model file drives the operator API (install() alone, no fast layers) - the call sequence
    of WindowAttention.forward, model/stratified_transformer.py:183-208, with the tensors the unmodified BasicLayer hands it
    (:312-319): int64 index_0 / index_1 / offsets, a 0-dim n_max tensor, fresh `.int()` copies for every operator, the rel-pos
    index evaluated with torch ops and range-checked with two host syncs (:186-190), `attn + bias`, scatter_softmax through the
    compat shim.  What this leaves out of the model's cost is its own torch index build (grid_sample / get_indice_pairs / sort /
    bincount, :10-65,:302-317: it lives in the model file and cannot run on the GPU box); scene_pass(fused="model") rebuilds the
    block's pattern per block with the package's device build instead - a LOWER bound of the model's index cost."""
    from . import compat
    q, k, v = state.q, state.k, state.v
    tq, tk, tv = state.tables
    xyz = state.xyz
    index_0, index_1, offsets, n_max = blk.index_0.long(), blk.index_1.long(), blk.offsets.long(), blk.n_max   # as :312-317 leaves them
    L = tq.shape[0]
    w, quant = (window_size, quant_size) if window_size is not None else state.window_quant

    def rel_index():
        rel = xyz[index_0] - xyz[index_1]                                      # :186
        rel = torch.round(rel * 100000) / 100000                               # :187
        idx = (rel + 2 * w - 0.0001) // quant                                  # :188
        # :189-190 are two asserts = two host syncs.  The syncs are paid here; the verdict is not enforced: the synthetic scenes have a
        # few pairs exactly at a window's extremes (index -1 or L), which every other leg clamps (index_build, the CPU port) - and
        # so does this one, after the syncs.
        bool((idx >= 0).all())                                                 # :189 (host sync)
        bool((idx <= L - 1).all())                                             # :190 (host sync)
        return idx.clamp_(0, L - 1)
    a1 = timer.run("attn_fwd/A1", P.attention_step1_v2, q.float(), k.float(), index_1.int(), offsets.int(), n_max)
    rel = timer.run("attn_fwd/rel_idx", rel_index)
    a2 = timer.run("attn_fwd/A2", P.dot_prod_with_idx_v3, q.float(), offsets.int(), n_max, k.float(), index_1.int(), tq.float(), tk.float(), rel.int())
    s = timer.run("attn_fwd/add", torch.add, a1, a2)
    sm = timer.run("attn_fwd/A3", compat.scatter_softmax, s, index_0, 0)
    out = timer.run("attn_fwd/A4", P.attention_step2_with_rel_pos_value_v2, sm.float(), v.float(), offsets.int(), n_max, index_1.int(), tv.float(), rel.int())
    timer.run("attn_bwd", out.backward, state.grad_out)
    return out


def _offsets_tensor(values, device):
    t = torch.tensor(values, dtype=torch.int32, device=device)
    P.hint_host_offsets(t, values)
    return t


_GEO_STREAMS = {}


def geometry_stream(device, which=0):
    """Side streams.  0: the sampling chain (FPS -> gather -> FPS ...), sequential by nature and one CU wide, so
    it runs BESIDE the attention blocks of the main stream instead of in front of them.  1: the kNN queries,
    which only consume the sampling chain's results and nobody on the chain waits for."""
    key = (torch.device(device).index, which)
    if key not in _GEO_STREAMS:
        # (a high-priority queue for the sampling chain was measured and does not help: 38.0 -> 41.9 ms per pass)
        _GEO_STREAMS[key] = torch.cuda.Stream(device=device)
    return _GEO_STREAMS[key]


EVEN_FIRST = os.environ.get("P2_EVEN_FIRST", "1") != "0"  # the first stage's first block is enqueued from inside its index build
SPECULATION = {"passes": 0, "reruns": 0}  # speculated passes / how many of them had to be run again (an exact tie among later-stage samples)
SPECULATE = os.environ.get("P2_SPECULATE", "1") != "0"  # later stages' samples are taken as the identity prefix while the sampler verifies them
INDEX_THREAD = os.environ.get("P2_INDEX_THREAD", "0") == "1"  # measured: 16.2 ms against 15.5 ms per pass (the two host threads contend), so opt-in


class _IndexBuilder:
    """Runs index(si) for every stage, in order, on a helper thread; wait(si) returns when stage si's index is enqueued
    (its tensors exist, its event is recorded) and re-raises what the build raised."""

    def __init__(self, index_fn, stages):
        import threading
        self._done = {si: threading.Event() for si in stages}
        self._error = None
        dev = torch.cuda.current_device()

        def work():
            try:
                torch.cuda.set_device(dev)
                for si in stages:
                    index_fn(si)
                    self._done[si].set()
            except BaseException as e:  # noqa: BLE001 - handed to the waiting thread
                self._error = e
            finally:
                for ev in self._done.values():
                    ev.set()

        self._thread = threading.Thread(target=work, name="p2-index-build", daemon=True)
        self._thread.start()

    def wait(self, si):
        self._done[si].wait()
        if self._error is not None:
            raise self._error

    def join(self):
        self._thread.join()
        if self._error is not None:
            raise self._error


_FLAG_HOST = {}  # (device index, lane) -> pinned bool [1]: the speculation check of a pass arrives here
_RETIRED = {}  # (device index, lane) -> deque of (event on the lane's main stream at the end of a pass, tensors kept alive until then)


def _reap(device, lane):
    import collections
    q = _RETIRED.setdefault((torch.device(device).index, lane), collections.deque())
    while q and (q[0][0].query() or len(q) > 4):
        if not q[0][0].query():
            q[0][0].synchronize()
        q.popleft()


def _retire(device, lane, main, keep):
    ev = torch.cuda.Event()
    ev.record(main)
    _RETIRED[(torch.device(device).index, lane)].append((ev, keep))


def _block_tensors(blk):
    """Every device tensor of a block pattern that the attention kernels read: allocated on the index stream, consumed on the main
    stream - all of them are recorded there (the pair list AND the cell plan: relp, cell_keys, ... - the caching allocator must not
    hand a plan's block to the next batch's index build while a block's kernels still read it)."""
    t = (blk.index_0, blk.index_1, blk.offsets, blk.rel_idx, blk.n_max)
    return t + (tuple(blk.cells.tensors()) if getattr(blk, "cells", None) is not None else ())


def scene_pass(xyz, offset, cfg, states=None, timer=None, seed=0, overlap=True, use_hip_index=True, fused=False, lane=0, cells=False, shard=None,
               speculate=None):
    """Runs the whole unit once (both phases of scene_pass_phases back to back).  Returns (states, results)."""
    gen = scene_pass_phases(xyz, offset, cfg, states, timer, seed, overlap, use_hip_index, fused, lane, cells=cells, shard=shard, speculate=speculate)
    next(gen)
    try:
        next(gen)
    except StopIteration as done:
        return done.value
    raise RuntimeError("scene_pass_phases yielded twice")


def scene_pass_phases(xyz, offset, cfg, states=None, timer=None, seed=0, overlap=True, use_hip_index=True, fused=False, lane=0,
                      inputs_resident=False, offset_host=None, cells=False, shard=None, speculate=None):
    """Generator form of scene_pass: the first next() enqueues the geometry chain of ALL stages (no host sync in
    it) and yields; the second runs the index builds (which stop the host: key width, pair count) and the attention
    blocks, and returns (states, results) through StopIteration.  passes_in_flight puts the first phase of the next
    batch in front of the second phase of this one.

    Runs the whole unit once.  `states` (list of StageState) carries the resident synthetic tensors;
    pass None on the first call to have them created (not timed by bench.py).  Returns (states, results).

    overlap=True: the geometry chain of all stages (stratified FPS, its continuation to the TransitionDown
    sample count, the gather of the next stage's points, kNN-16, kNN-3) is enqueued on a side stream; the
    main stream (index build + attention forward/backward) only waits for the first n//scale+1 samples of
    its own stage.  Nothing is skipped and every dependency is an event; results are identical.

    lane: which set of side streams the pass uses.  Two passes over different batches that are enqueued under
    different current streams, with different lanes and different `states`, share no stream and no tensor they
    write, so the device may run them side by side (passes_in_flight): the sampling chain of the next batch - one
    CU wide, and a function of the coordinates alone - then runs beside the attention blocks of this one.

    speculate (default: on for a pass on its own with overlap, P2_SPECULATE=0 turns it off): every cloud after the first is the output
    of an FPS kept in selection order (TransitionDown, :103-104), and FPS of such a cloud returns 0, 1, 2, ... unless two candidates
    tie exactly (csrc/fps_bucket.hip, identity-prefix verification).  The index builds, the next clouds and the kNN queries of those
    stages therefore take the identity prefix AT ONCE, the sampler - which verifies exactly that prefix and samples on where it
    fails - runs beside them on the geometry stream, and the pass compares the two at its end: equal (always, ties aside) means
    every tensor of the pass is what the unspeculated pass computes; unequal, the pass is run again without speculation.  Nothing
    is skipped - the samplers run in full - they just no longer stand between one stage's blocks and the next's."""
    timer = timer or Timer(False)
    if speculate is None:
        speculate = SPECULATE and overlap and not inputs_resident and use_hip_index
    call_args = (xyz, offset, cfg, states, timer, seed, overlap, use_hip_index, fused, lane, inputs_resident, offset_host, cells, shard)
    # Tensors that are allocated under one stream and read by kernels of another (index tensors and cell plans: index stream ->
    # main; samples, clouds, offsets: geometry / upload streams -> the others) are kept alive in `keep` until an event recorded
    # at the END of the pass on the main stream has completed (_retire), instead of Tensor.record_stream(): that makes the
    # allocator record one event per tensor and stream when the tensor is freed - ~200 markers at the end of a pass, 0.6-0.7 ms
    # of the main stream between two passes (tools/span_timeline.py, round 3).
    keep = []
    _reap(xyz.device, lane)
    # nothing is carried over from an earlier pass: the CSC transpositions and the FPS sampler state are
    # rebuilt inside every pass (they are reused only WITHIN a pass, between blocks / the two FPS calls of a stage)
    P.clear_caches()
    dev = xyz.device
    main = torch.cuda.current_stream(dev)
    geo = geometry_stream(dev, 3 * lane) if overlap else main
    knn_s = geometry_stream(dev, 3 * lane + 1) if overlap else main
    if overlap and not inputs_resident:
        # xyz / offset may have been produced on the current stream just now.  (inputs_resident: the caller vouches
        # that they are complete - passes_in_flight - and the side streams need not wait for whatever the current
        # stream still holds, i.e. the attention blocks of the lane's previous batch.)
        geo.wait_stream(main)
        knn_s.wait_stream(main)
        geometry_stream(dev, 3 * lane + 2).wait_stream(main)
    # offset_host: the caller's host copy of `offset` (a data loader has it).  Without it the offsets are read back,
    # which synchronises the current stream - with batches in flight that means waiting for the attention blocks of
    # the lane's previous batch.
    if offset_host is None:
        offset_host = offset.tolist()
    offset_host = [int(o) for o in offset_host]
    P.hint_host_offsets(offset, offset_host)
    make = states is None
    states = [] if make else states
    results = []
    first = 0 if cfg.stem_transformer else 1

    def on_geo():
        return torch.cuda.stream(geo)

    # Every stage's offsets follow from the host-side counts alone (transition_down_offset / stratified_new_offset
    # are integer rules on the batch sizes), so all the small offset tensors are uploaded NOW, while both streams
    # are empty: a torch.tensor(..., device=...) issued later would queue its host-to-device copy behind the
    # sampler on the geometry stream and block the host - and with it the launches of the next stages - for
    # the ~30 ms the stage-0 sampling takes.
    plan_host = [offset_host]
    n_trans = (len(cfg.stages) - first - 1) + (0 if cfg.stem_transformer else 1)
    for _ in range(n_trans):
        plan_host.append(index_build.transition_down_offset(plan_host[-1], cfg.ratio))
    # They go through the index stream, which is idle by construction (the host has read every result of the lane's
    # previous batch from it): a host-to-device copy from pageable memory blocks the host until the stream it was
    # queued on reaches it; the lane's main stream may still hold the previous batch's blocks, its kNN stream the
    # upsampling queries behind the whole sampling chain.  (Not a further stream: the runtime multiplexes streams onto
    # hardware queues, and an upload stream that lands in a sampler's queue stops the host for the sampler's 27 ms.)
    up_s = geometry_stream(dev, 3 * lane + 2) if overlap else main
    with torch.cuda.stream(up_s):
        # ONE host-to-device copy for all of them (seven small torch.tensor(..., device=) uploads from pageable memory cost ~50 us each,
        # ~0.35 ms in front of the stage-0 sampler, i.e. on the pass's critical path); the per-stage tensors are views of it
        strat_host = [index_build.stratified_new_offset(v, cfg.downsample_scale) for v in plan_host]
        rows = plan_host[1:] + strat_host
        flat = torch.tensor([x for r_ in rows for x in r_], dtype=torch.int32).to(dev)
        views, at = [], 0
        for r_ in rows:
            t_ = flat[at:at + len(r_)]
            P.hint_host_offsets(t_, r_)
            views.append(t_)
            at += len(r_)
        plan_dev = [offset] + views[:len(plan_host) - 1]
        strat_dev = views[len(plan_host) - 1:]
    if overlap:
        for s_ in (geo, knn_s, main):
            s_.wait_stream(up_s)
            keep.extend(plan_dev[1:] + strat_dev)
    level = [0]

    checks = []      # 0-d bool tensors (geometry stream): a sampler's output differs from the identity prefix that was used in its place
    ordered = set()  # ids of clouds that are an FPS output in selection order
    guesses = {}     # id of such a cloud -> the identity prefix its stratified sampling is expected to return (exists with the cloud)

    def prefix(off_host, new_off_host):
        """the identity prefix of every batch element: rows start_b ... start_b + count_b - 1"""
        if len(off_host) == 1:
            return torch.arange(new_off_host[0], dtype=torch.int32, device=dev)
        starts = [0] + list(off_host[:-1])
        counts = [new_off_host[0]] + [new_off_host[i] - new_off_host[i - 1] for i in range(1, len(new_off_host))]
        return torch.cat([torch.arange(s_, s_ + c_, dtype=torch.int32, device=dev) for s_, c_ in zip(starts, counts)])

    def transition(x, off, off_host):
        """TransitionDown's sampling + grouping indices (:98-106), on the geometry stream"""
        level[0] += 1
        n_off_host, n_offset = plan_host[level[0]], plan_dev[level[0]]
        if speculate and id(x) in ordered:
            # the next cloud is the identity prefix of this one: it exists NOW (kNN stream), the sampler verifies it beside
            with torch.cuda.stream(knn_s):
                pred = prefix(off_host, n_off_host)
                n_xyz = x[:n_off_host[0]] if len(off_host) == 1 else x[pred.long(), :].contiguous()
                guesses[id(n_xyz)] = prefix(n_off_host, strat_host[level[0]])
                ready = torch.cuda.Event()
                ready.record(knn_s)
                cloud_ready[level[0]] = ready
                knn_idx, _ = timer.run("knn/k16", P.knnquery, cfg.k, x, n_xyz, off, n_offset)
            geo.wait_event(ready)  # (pred)
            idx = timer.run("fps/transition", P.furthestsampling, x, off, n_offset)
            checks.append((idx != pred).any())
            ordered.add(id(n_xyz))
            keep.extend((x, n_xyz, off, n_offset, pred, idx))
            return n_xyz, n_offset, n_off_host, knn_idx
        idx = timer.run("fps/transition", P.furthestsampling, x, off, n_offset)
        n_xyz = x[idx.long(), :].contiguous()
        if speculate:
            guesses[id(n_xyz)] = prefix(n_off_host, strat_host[level[0]])
        ready = torch.cuda.Event()
        ready.record(geo)
        cloud_ready[level[0]] = ready
        if overlap:  # the grouping query leaves the sampling chain here
            knn_s.wait_stream(geo)
            keep.extend((x, n_xyz, off, n_offset))
        ordered.add(id(n_xyz))
        with torch.cuda.stream(knn_s):
            knn_idx, _ = timer.run("knn/k16", P.knnquery, cfg.k, x, n_xyz, off, n_offset)
        return n_xyz, n_offset, n_off_host, knn_idx

    cloud_ready = {}  # transition level -> event: the level's point cloud exists (recorded on the geometry stream)
    cur_xyz, cur_off, cur_off_host = xyz, offset, offset_host
    P.hint_unordered(xyz)  # (a raw scene: no identity-prefix probe in front of its sampler)
    if not cfg.stem_transformer:  # Stratified.forward :458-462: a TransitionDown precedes the first attention stage
        with on_geo():
            cur_xyz, cur_off, cur_off_host, _ = transition(cur_xyz, cur_off, cur_off_host)
    # The loop is software-pipelined on the HOST: while the attention blocks of stage s run on the main stream,
    # the geometry of stage s+1 is already queued on the geometry stream and its index build - whose two host
    # syncs (key width, pair count) would otherwise stop the launches - is done on a third stream.
    idx_s = geometry_stream(dev, 3 * lane + 2) if overlap else main
    stages = list(range(first, len(cfg.stages)))
    clouds = {}   # si -> (xyz, off, off_host)
    geo_out = {}  # si -> (ds, ev_ds, knn_idx or None)
    idx_out = {}  # si -> (even, odd, ev_idx)
    clouds[first] = (cur_xyz, cur_off, cur_off_host)

    def geometry(si):
        """geometry stream: samples for this stage's stratified keys, then on to the next stage's points"""
        x, off, off_host = clouds[si]
        with on_geo():
            guess = guesses.get(id(x)) if (speculate and id(x) in ordered) else None
            if guess is not None and level[0] in cloud_ready:
                geo.wait_event(cloud_ready[level[0]])
            ds = timer.run("fps/stratified", P.furthestsampling, x, off, strat_dev[level[0]])
            if guess is not None:
                checks.append((ds != guess).any())
                ev_ds = None  # the index build takes `guess` and waits for the cloud alone
            else:
                ev_ds = torch.cuda.Event()
                ev_ds.record(geo)
            knn_idx = None
            if si < len(cfg.stages) - 1:
                n_xyz, n_off, n_off_host, knn_idx = transition(x, off, off_host)
                clouds[si + 1] = (n_xyz, n_off, n_off_host)
        geo_out[si] = (ds if guess is None else guess, ev_ds, knn_idx, ds)

    def index(si, on_even=None):
        x, off, _ = clouds[si]
        ds, ev_ds, _, ds_sampled = geo_out[si]
        st = cfg.stages[si]
        if overlap:
            keep.extend((ds, ds_sampled, x, off))
        with torch.cuda.stream(idx_s):
            parts_ctx = None
            if use_hip_index:
                # the window partitions need the coordinates only: they run BESIDE the stage's sampling (which the pair
                # lists below have to wait for), not behind it
                lvl = si - first + (0 if cfg.stem_transformer else 1)
                if overlap and lvl in cloud_ready:
                    idx_s.wait_event(cloud_ready[lvl])
                want_cells = cells or str(fused).startswith("cell")
                parts_ctx = timer.run("index/partitions", index_build.stage_partitions_hip, x, off, st.window_size, None,
                                      index_build.cell_query_cap(x.shape[0], st.num_heads) if want_cells else None)
            if overlap and ev_ds is not None:
                idx_s.wait_event(ev_ds)
            if use_hip_index:
                even, odd, _ = timer.run("index/build", index_build.stage_index_hip, x, off, st.window_size, st.quant_size, ds,
                                         table_rows(st) if (cells or str(fused).startswith("cell")) else None,
                                         index_build.cell_query_cap(x.shape[0], st.num_heads), parts_ctx, on_even)
            else:
                parts = timer.run("index/partition", index_build.stage_partitions, x, off, st.window_size)
                even = timer.run("index/pairs", index_build.build_block_index, x, parts["small"], parts["large"], ds, st.window_size, st.quant_size, False)
                odd = timer.run("index/pairs", index_build.build_block_index, x, parts["small_shift"], parts["large_shift"], ds, st.window_size, st.quant_size, True)
            ev_idx = torch.cuda.Event()
            ev_idx.record(idx_s)
        idx_out[si] = (even, odd, ev_idx, parts_ctx)

    for si in stages:
        geometry(si)
    # Upsample chain (:479-480): interpolate from the coarse stage back to each finer one (kNN k=up_k)
    if overlap:
        knn_s.wait_stream(geo)
    with torch.cuda.stream(knn_s):
        for fine in reversed(stages[:-1]):
            cx, coff, _ = clouds[fine + 1]
            fx, foff, _ = clouds[fine]
            timer.run("knn/k3", P.knnquery, cfg.up_k, cx, fx, coff, foff)
    after = yield "geometry queued"
    if after is not None:  # passes_in_flight: the attention blocks of this batch follow those of the previous one
        main.wait_event(after)
    # The index builds stop the host twice each (key width, pair count) and each waits for its stage's samples.  They run on a
    # helper thread (own stream; the library's launch state is thread-local), one stage after the other as the geometry
    # arrives, so the launches of the attention blocks never wait behind a later stage's sampling.
    builder = _IndexBuilder(index, stages) if (overlap and INDEX_THREAD) else None
    started = {}  # si -> output of block 0, when it was enqueued from inside the stage's index build

    def first_block_early(even_blk):
        """Called by the first stage's index build when the plain pattern exists: its block runs beside the build of the shifted
        pattern (the pass's start is sampler -> index build -> first block; this takes the second pattern off that path)."""
        si = first
        st = cfg.stages[si]
        x, off, _ = clouds[si]
        ds = geo_out[si][0]
        cells_only = str(fused).startswith("cell") and shard is None and even_blk.cells_ready is not None
        if cells_only:  # the window-centric kernels need the plan alone: they do not wait for the pair-list tensors
            ev = even_blk.cells_ready
        else:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))  # (the index stream)
        with torch.cuda.stream(main):
            main.wait_event(ev)
            keep.extend(t for t in (x, off, ds) + _block_tensors(even_blk) if torch.is_tensor(t))
            if make:
                states.append(make_stage_state(x, off, st, seed + si))
            state = states[si - first]
            state.xyz, state.offset = x, off
            timer.stage = si
            started[si] = attention_block(state, even_blk, timer, fused, shard)
            timer.stage = None

    halo_mode = shard is not None and len(shard) > 2 and shard[2] == "halo"  # (ownership is cut over BOTH patterns' pairs: no early block)
    if builder is None:
        index(first, first_block_early if (overlap and use_hip_index and EVEN_FIRST and not halo_mode) else None)
    for si in stages:
        st = cfg.stages[si]
        x, off, _ = clouds[si]
        if builder is not None:
            builder.wait(si)
        even, odd, ev_idx, parts_ctx = idx_out[si]
        if shard is not None and len(shard) > 2 and shard[2] == "halo" and even is not None and odd is not None:
            from . import sharding as _sh
            if even.owner is None:
                even.owner = _sh.window_owners(even.parts["large"], [even.offsets, odd.offsets], shard[1])
            odd.owner = even.owner   # one ownership per stage: the rows do not move between the blocks
        _, _, knn_idx, ds = geo_out[si]
        if overlap:
            cells_only = str(fused).startswith("cell") and shard is None and odd is not None and odd.cells_ready is not None
            main.wait_event(odd.cells_ready if cells_only else ev_idx)  # (same stream: the shifted pattern's plan is the later one)
            keep.extend(t for t in (x, off, ds) + _block_tensors(even) + _block_tensors(odd) if torch.is_tensor(t))
        if make and si not in started:
            states.append(make_stage_state(x, off, st, seed + si))
        state = states[si - first]
        state.xyz, state.offset = x, off
        out = started.get(si)
        # the next stage's index build stops the host twice; it is issued where the host can afford to wait:
        # behind ALL blocks of the first stage (its samples only arrive when the stage-0 sampler is through),
        # behind the FIRST block of a later stage (its samples are long there, and the late stages' blocks are
        # short enough for the launches to fall behind otherwise)
        # (a two-block stage enqueues both blocks first: the host would otherwise sit in the next stage's index syncs while the device
        #  has nothing of this stage left - a 0.4 ms hole between the two blocks of stage 1, tools/span_timeline.py)
        early = si + 1 in stages and si > first
        # A pass on its own: behind block min(depth - 1, 3), so that the device has blocks queued while the host sits in the syncs (issued
        # behind block 0, the host left a 0.3-0.4 ms hole in stages 1 and 2, tools/span_timeline.py).  Batches in flight keep round 2's
        # order (behind block 0): there the host is the limit and the later position measured 11.0 -> 15.5 ms per batch.
        early_after = 0 if inputs_resident else min(st.depth - 1, 3)
        for b in range(st.depth):
            if b == 0 and si in started:
                continue
            blk_b = even if b % 2 == 0 else odd
            if fused == "model" and use_hip_index and b >= 2:
                # the unmodified BasicLayer rebuilds the block's pair list for EVERY block (:302-317); blocks 0 and 1 use the
                # stage's build above, every further block pays one pattern's build again (on the main stream, as the model would)
                rebuilt = timer.run("index/rebuild", index_build.stage_index_hip, x, off, st.window_size, st.quant_size, ds, None, 0,
                                    parts_ctx, None, (b % 2,))
                blk_b = rebuilt[b % 2]
            timer.stage = si
            out = attention_block(state, blk_b, timer, fused, shard)
            timer.stage = None
            if early and b == early_after and builder is None:
                index(si + 1)
        results.append(dict(stage=si, n=x.shape[0], M_even=int(even.index_1.shape[0]), M_odd=int(odd.index_1.shape[0]),
                            even=even, odd=odd, downsample_idx=ds, out=out))
        if shard is not None:  # `out` holds the rank's rows of the last block only
            last_blk = even if (st.depth - 1) % 2 == 0 else odd
            if len(shard) > 2 and shard[2] == "halo":
                results[-1]["out_ids"] = last_blk.halo_rows
                results[-1]["halo_fraction"] = [(b_.halo[0][1] if isinstance(b_.halo[0], tuple) else b_.halo[0].halo).halo_fraction() for b_ in (even, odd)]
            else:
                results[-1]["out_rows"] = (last_blk.shard[0].lo, last_blk.shard[0].hi)
        if knn_idx is not None:
            results[-1]["transition_knn"] = knn_idx
        if si + 1 in stages and not early and builder is None:
            index(si + 1)
    if builder is not None:
        builder.join()
    timer.run("mark/blocks_done", lambda: None)
    if overlap:
        main.wait_stream(geo)
        main.wait_stream(knn_s)
        main.wait_stream(idx_s)
    timer.run("mark/streams_joined", lambda: None)
    if checks:
        with on_geo():
            # (one pinned word per lane, reused: a fresh pinned allocation per pass showed as 3-5 ms outliers, tools/pass_times.py)
            wrong = _FLAG_HOST.get((dev.index, lane))
            if wrong is None:
                wrong = _FLAG_HOST[(dev.index, lane)] = torch.empty(1, dtype=torch.bool, pin_memory=True)
            wrong.copy_(torch.stack(checks).any().reshape(1), non_blocking=True)
            seen = torch.cuda.Event()
            seen.record(geo)
        keep.extend(checks)
    _retire(dev, lane, main, keep)
    if checks:
        SPECULATION["passes"] += 1
        seen.synchronize()  # the geometry stream has long finished: the blocks of the last stages are what the device still holds
        if bool(wrong):
            # an exact tie among the samples of a later stage: the identity prefix was not what the sampler returns - once more, waiting
            # for every sampler as the reference does (same states: the synthetic tensors are inputs, the gradients are reset per block)
            SPECULATION["reruns"] += 1
            torch.cuda.synchronize(dev)
            a = call_args
            return scene_pass(a[0], a[1], a[2], states, a[4], a[5], a[6], a[7], a[8], a[9], cells=a[12], shard=a[13], speculate=False)
    return states, results


def passes_in_flight(xyz_list, offset_list, cfg, lanes, steps, timer=None, fused=False, offset_host_list=None, paced=True):
    """`steps` passes, batch k on lane k % len(lanes); lanes = [(stream, states), ...], one set of resident state
    tensors per lane.  The geometry chains of the next len(lanes)-1 batches are queued in front of the index builds
    and attention blocks of batch k (scene_pass_phases); nothing is shared between two lanes but the read-only inputs, which have to be complete (resident) when this is called.  The caller synchronizes the device before and after.
    Returns the results of the last pass of every lane."""
    last = [None] * len(lanes)

    def start(k):
        stream, states = lanes[k % len(lanes)]
        with torch.cuda.stream(stream):
            gen = scene_pass_phases(xyz_list[k % len(xyz_list)], offset_list[k % len(offset_list)], cfg, states, timer,
                                    fused=fused, lane=k % len(lanes), inputs_resident=True,
                                    offset_host=None if offset_host_list is None else offset_host_list[k % len(offset_host_list)])
            next(gen)  # the geometry chain of batch k is queued
        return gen

    prev_done = None
    done_events = []
    pace_lag = int(os.environ.get("P2_PACE_LAG", "1"))  # released on batch k-lag: 22.2 / 24.0 / 28.8 ms per step for 1 / 2 / 3
    ahead = len(lanes) - 1  # geometry phases queued in front of the attention phase being enqueued
    queue = [start(k) for k in range(min(ahead, steps))]
    blocks_first = os.environ.get("P2_BLOCKS_FIRST", "1") != "0" and ahead > 0

    def release_geometry(k):
        if k + ahead < steps:
            # Pacing: the geometry of batch k+ahead is released when batch k-1 is through, so a fixed number of
            # sampling chains is in flight, evenly spaced.  Released as early as the host can (three chains start
            # in a burst, slow each other and the attention kernels beside them) a step takes 28.5 instead of 24.8 ms;
            # with the same dependency on the device only (side streams waiting for the lane's main stream) 26.9 ms.
            if paced and len(done_events) >= pace_lag + (1 if blocks_first else 0):
                done_events[-pace_lag - (1 if blocks_first else 0)].synchronize()
            queue.append(start(k + ahead))

    for k in range(steps):
        # Round 2: the index builds and blocks of batch k are enqueued BEFORE the host waits for batch k-1 (the pacing of the
        # geometry): with the sampler at 5 instead of 27 ms the wait, the geometry launches and the first index build of batch k
        # otherwise leave the attention stream idle for 2-3 ms per batch (P2_BLOCKS_FIRST=0: the round-1 order).
        if not blocks_first:
            release_geometry(k)
        gen = queue.pop(0)
        with torch.cuda.stream(lanes[k % len(lanes)][0]):
            try:
                gen.send(prev_done)
                raise RuntimeError("scene_pass_phases yielded twice")
            except StopIteration as done:
                _, last[k % len(lanes)] = done.value
            # the forward+backward of batch k+1 may not start before that of batch k is through (in a training loop
            # the optimizer step sits between them): only the data-side work - sampling, kNN - runs ahead
            prev_done = torch.cuda.Event()
            prev_done.record(torch.cuda.current_stream())
            done_events.append(prev_done)
        if blocks_first:
            release_geometry(k)
    return last
