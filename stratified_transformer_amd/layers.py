"""Installable fast forms of the model methods that drive the hot path, with the reference's own signatures:

    BasicLayer.forward(self, feats, xyz, offset)                                   model/stratified_transformer.py:267-326
    WindowAttention.forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max)            ...:164-217
    TransitionDown.forward(self, feats, xyz, offset)                                                ...:96-111

`install_fast_layers()` (or `stratified_transformer_amd.install(fast_layers=True)`) rebinds these methods on the classes of
the UNMODIFIED model file; the modules' own parameters are used as they are (`qkv`, `proj`, the three rel-pos tables, `norm1`,
`mlp`, `downsample`, ...), so checkpoints, optimizers and the rest of the model (`SwinTransformerBlock.forward`, `TransitionDown`,
`Upsample`, stem, classifier) are untouched.  What changes is WHERE the index work and the attention run:

  * BasicLayer.forward builds the stage's index ONCE (the reference rebuilds the pair list for every block, :302-317, although
    only two patterns exist per stage): FPS through the operator API (bit-exact, resumed later by TransitionDown's call), then
    `index_build.stage_index_hip` -> even / odd pair lists, rel-pos indices and cell plans, on the device, two host syncs per
    stage instead of the reference's ~10 + 2 per block (:60, :283-287, boolean-mask gathers, the asserts of :189-190).
  * every block is still called through the reference's own `SwinTransformerBlock.forward` (`blk(feats, xyz, index_0, index_1,
    index_0_offsets, n_max)`, :319) with the pattern's index tensors; the pattern's cell plan travels beside them on the block's
    attention module (`_sta_block`).
  * WindowAttention.forward computes qkv / scale / proj exactly as :180-182,212-215 and replaces :183-208 (three operators, the
    `+`, scatter_softmax, two range asserts) by ONE autograd function: `fused.cell_attention` on the plan when BasicLayer handed
    one over (d = 16, L <= 80 - every shipped config), else `fused.window_attention` on the pair list it was given (d = 16),
    else the five operators in the reference's order.  Called on its own (no plan), it needs nothing but the reference's
    arguments.

  * The geometry chain (CHAIN, on by default): BasicLayer.forward puts its stage's samplers (stratified FPS, its continuation to
    TransitionDown's count), the next cloud and TransitionDown's grouping query on side streams, beside its attention blocks;
    TransitionDown.forward takes them from there (its grouping / norm / linear / max-pool are the module's own) and hands the
    next BasicLayer a cloud the layers know to be in selection order - whose samples are then taken as the identity prefix at
    once while the sampler verifies them; a stage whose check fails (an exact tie) runs again with the sampler's answer.  This
    is pipeline.scene_pass's schedule under the unmodified model's call order.  With only BasicLayer / WindowAttention rebound,
    the model's own TransitionDown finds the samples in the sampler's kept state (ordered behind the side stream by an event).

Numbers: the same sums in another order (tests/test_hip_parity.py::test_installed_fast_layers_against_the_reference_layer:
output and every parameter gradient of a depth-2 BasicLayer with TransitionDown against the reference's own run, <= 1e-3).
"""
import os
import sys
import weakref

import torch

from . import fused, index_build
from . import pointops as P

_ORIGINAL = {}

# The geometry chain (round 3): the installed BasicLayer.forward puts the samplers and the TransitionDown geometry of its stage on side
# streams, beside its attention blocks, and the installed TransitionDown.forward picks them up - the schedule of pipeline.scene_pass
# under the unmodified model's own call order.  P2_LAYER_CHAIN=0: everything on the caller's stream, in the model's order.
CHAIN = os.environ.get("P2_LAYER_CHAIN", "1") != "0"
SPECULATE = os.environ.get("P2_SPECULATE", "1") != "0"
STATS = {"layers": 0, "speculated": 0, "reruns": 0, "transitions_prefetched": 0}
_CLOUDS = {}  # id(xyz) -> _Cloud
_FLAG_HOST = {}  # device index -> ring of pinned bool [1] words (speculation checks)


class _Cloud:
    """What the installed layers know about one point cloud of a forward pass."""
    __slots__ = ("ref", "off_host", "ordered", "trans", "__weakref__")

    def __init__(self, xyz, off_host, ordered):
        self.ref = weakref.ref(xyz)
        self.off_host = off_host      # host copy of the offsets (no read-back)
        self.ordered = ordered        # the cloud is an FPS output in selection order (TransitionDown, :103-104)
        self.trans = None             # the TransitionDown geometry of this cloud, prefetched: dict(ratio, k, n_offset, n_xyz, knn, ready)


def _cloud_of(xyz):
    c = _CLOUDS.get(id(xyz))
    return c if c is not None and c.ref() is xyz else None


def _remember(xyz, cloud):
    key = id(xyz)
    _CLOUDS[key] = cloud
    weakref.finalize(xyz, lambda k=key, c=cloud: _CLOUDS.pop(k, None) if _CLOUDS.get(k) is c else None)
    return cloud


def forget_clouds():
    """Drops what the layers remember about the clouds they have seen (a NEW batch is a new tensor and is never found; a caller that
    feeds the same tensor object again - a benchmark loop - calls this, with pointops.clear_caches(), to have the geometry recomputed)."""
    _CLOUDS.clear()


def _prefix(off_host, new_off_host, dev):
    """rows start_b ... start_b + count_b - 1 of every batch element: what FPS returns on a cloud in selection order (ties aside)"""
    if len(off_host) == 1:
        return torch.arange(new_off_host[0], dtype=torch.int32, device=dev)
    starts = [0] + list(off_host[:-1])
    counts = [new_off_host[0]] + [new_off_host[i] - new_off_host[i - 1] for i in range(1, len(new_off_host))]
    return torch.cat([torch.arange(s_, s_ + c_, dtype=torch.int32, device=dev) for s_, c_ in zip(starts, counts)])


_STAGING = {}  # device index -> [pinned int32 ring [slots, width], next slot]


def _offsets(values, dev):
    """Offsets known on the host -> device tensor, without stopping the host: torch.tensor(..., device=) copies from pageable memory
    (~0.3 ms of host time each, tools/host_profile_layers.py); a slot of a small pinned ring + a non-blocking copy does not.  A slot is
    written again 64 uploads later - several forward passes, each of which reads results back (the index build) after its uploads."""
    n = len(values)
    if n > 64:
        t = torch.tensor(values, dtype=torch.int32, device=dev)
    else:
        ring = _STAGING.get(dev.index)
        if ring is None:
            ring = _STAGING[dev.index] = [torch.empty((64, 64), dtype=torch.int32, pin_memory=True), 0]
        slot = ring[0][ring[1] % 64]
        ring[1] += 1
        slot[:n] = torch.tensor(values, dtype=torch.int32)
        t = slot[:n].to(dev, non_blocking=True)
    P.hint_host_offsets(t, values)
    return t


def _original(obj):
    for cls in type(obj).__mro__:
        if (cls, "forward") in _ORIGINAL:
            return _ORIGINAL[cls, "forward"]
    raise RuntimeError("fast layers: the original forward of %s is not known" % type(obj).__name__)


def _heads_dim(attn):
    h = int(attn.num_heads)
    return h, int(attn.dim) // h


def window_attention_forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max):
    """Replacement of WindowAttention.forward (:164-217), same arguments, same result."""
    N, C = feats.shape
    h, d = _heads_dim(self)
    assert index_0.shape[0] == index_1.shape[0]
    if not (self.rel_query and self.rel_key and self.rel_value) or not feats.is_cuda:
        return _original(self)(self, feats, xyz, index_0, index_1, index_0_offsets, n_max)
    qkv = self.qkv(feats).reshape(N, 3, h, d).permute(1, 0, 2, 3).contiguous()          # :180
    query, key, value = qkv[0], qkv[1], qkv[2]
    query = (query * self.scale).float()                                                  # :182-183 (`.float()`: autocast)
    key, value = key.float(), value.float()
    tq, tk, tv = (t.float().contiguous() for t in (self.relative_pos_query_table, self.relative_pos_key_table, self.relative_pos_value_table))
    L = int(tq.shape[0])
    blk = getattr(self, "_sta_block", None)
    plan = getattr(blk, "cells", None) if blk is not None else None
    if plan is not None and d == 16 and L <= 80 and plan.table_rows == L and plan.n_points == N:
        x = fused.cell_attention(query, key, value, tq, tk, tv, plan)
    else:
        offs, i1 = index_0_offsets.int().contiguous(), index_1.int().contiguous()
        if blk is not None and blk.rel_idx is not None and blk.rel_idx.shape[0] == i1.shape[0]:
            rel = blk.rel_idx
        else:  # :186-188 with the GPU's arithmetic; the range asserts of :189-190 become a clamp (no host sync)
            rel = index_build.rel_pos_index(xyz, index_0, index_1, float(self.window_size), float(self.quant_size)).clamp_(0, L - 1).contiguous()
        if d == 16:
            x = fused.window_attention(query, key, value, tq, tk, tv, offs, i1, rel)
        else:
            a = P.attention_step1_v2(query, key, i1, offs, n_max) + P.dot_prod_with_idx_v3(query, offs, n_max, key, i1, tq, tk, rel)
            x = P.attention_step2_with_rel_pos_value_v2(P.segment_softmax(a, offs), value, offs, n_max, i1, tv, rel)
    x = x.view(N, C)
    return self.proj_drop(self.proj(x))                                                   # :212-215 (under autocast the Linear casts)


def _plain_layer_forward(self, feats, xyz, offset, attn0):
    """everything on the caller's stream, in the model's order"""
    h, d = _heads_dim(attn0)
    N = xyz.shape[0]
    xyz_c = xyz.float().contiguous()
    offset_i = offset.int().contiguous()
    off_host = offset_i.tolist()                                     # one host sync (the reference: one .item() per element, :283-287)
    P.hint_host_offsets(offset_i, off_host)
    new_host = index_build.stratified_new_offset(off_host, int(self.downsample_scale))
    new_offset = _offsets(new_host, xyz.device)
    downsample_idx = P.furthestsampling(xyz_c, offset_i, new_offset)                      # :289
    feats = _blocks(self, feats, xyz, xyz_c, offset_i, downsample_idx, attn0, h, d, N)
    if self.downsample:                                                                   # :321-324
        feats_down, xyz_down, offset_down = self.downsample(feats, xyz, offset)
    else:
        feats_down, xyz_down, offset_down = None, None, None
    return feats, xyz, offset, feats_down, xyz_down, offset_down


def _blocks(self, feats, xyz, xyz_c, offset_i, downsample_idx, attn0, h, d, N):
    L = int(attn0.relative_pos_query_table.shape[0])
    want_cells = d == 16 and L <= 80
    even, odd, _ = index_build.stage_index_hip(xyz_c, offset_i, float(self.window_size), float(attn0.quant_size), downsample_idx,
                                               cell_table_rows=L if want_cells else None,
                                               cell_max_queries=index_build.cell_query_cap(N, h) if want_cells else 0)
    for i, blk in enumerate(self.blocks):                                                 # :302-319
        bi = even if i % 2 == 0 else odd
        blk.attn._sta_block = bi
        try:
            feats = blk(feats, xyz, bi.index_0, bi.index_1, bi.offsets, bi.n_max)
        finally:
            blk.attn._sta_block = None
    return feats


def basic_layer_forward(self, feats, xyz, offset):
    """Replacement of BasicLayer.forward (:267-326), same arguments, same six results."""
    attn0 = self.blocks[0].attn
    if not (feats.is_cuda and attn0.rel_query and attn0.rel_key and attn0.rel_value):
        return _original(self)(self, feats, xyz, offset)
    STATS["layers"] += 1
    down = self.downsample if self.downsample else None
    chain = CHAIN and xyz.dtype == torch.float32 and xyz.is_contiguous() and offset.dtype == torch.int32 and offset.is_contiguous()
    if not chain:
        return _plain_layer_forward(self, feats, xyz, offset, attn0)
    # ---- the stage's samplers and the TransitionDown geometry on side streams, beside the blocks (pipeline.scene_pass's schedule) ----
    from .pipeline import geometry_stream
    h, d = _heads_dim(attn0)
    N, dev = xyz.shape[0], xyz.device
    cloud = _cloud_of(xyz)
    if cloud is None:
        off_host = P._host_list(offset)
        if off_host is None:
            off_host = offset.tolist()                               # one host sync (the reference: one .item() per element, :283-287)
            P.hint_host_offsets(offset, off_host)
        cloud = _remember(xyz, _Cloud(xyz, off_host, False))
    off_host = cloud.off_host
    main = torch.cuda.current_stream(dev)
    geo, knn_s = geometry_stream(dev, 0), geometry_stream(dev, 1)
    new_host = index_build.stratified_new_offset(off_host, int(self.downsample_scale))
    new_offset = _offsets(new_host, dev)
    guess_on = SPECULATE and cloud.ordered
    guess = _prefix(off_host, new_host, dev) if guess_on else None
    want_trans = down is not None and hasattr(down, "ratio") and hasattr(down, "k") and (cloud.trans is None or
                                                                                          (cloud.trans["ratio"], cloud.trans["k"]) != (down.ratio, down.k))
    t_host = t_off = t_guess = None
    if want_trans:
        t_host = index_build.transition_down_offset(off_host, down.ratio)
        t_off = _offsets(t_host, dev)
        t_guess = _prefix(off_host, t_host, dev) if guess_on else None
    start = torch.cuda.Event()
    start.record(main)
    checks = []
    geo.wait_event(start)
    with torch.cuda.stream(geo):
        sampled = P.furthestsampling(xyz, offset, new_offset)                             # :289 (on an ordered cloud: the verification)
        have_samples = torch.cuda.Event()
        have_samples.record(geo)
        if guess is not None:
            checks.append((sampled != guess).any())
    trans = None
    if want_trans:
        def grouping(n_xyz, after):
            knn_s.wait_event(after)
            with torch.cuda.stream(knn_s):
                knn_idx, _ = P.knnquery(int(down.k), xyz, n_xyz, offset, t_off)
                ready = torch.cuda.Event()
                ready.record(knn_s)
            return dict(ratio=down.ratio, k=down.k, n_offset=t_off, n_off_host=t_host, n_xyz=n_xyz, knn=knn_idx, ready=ready)

        if t_guess is not None:
            # the next cloud is the identity prefix of this one: it and its grouping query exist at once, the sampler verifies beside
            with torch.cuda.stream(knn_s):
                knn_s.wait_event(start)
                n_xyz = xyz[:t_host[0]] if len(off_host) == 1 else xyz[t_guess.long(), :].contiguous()
                made = torch.cuda.Event()
                made.record(knn_s)
            trans = grouping(n_xyz, made)
            with torch.cuda.stream(geo):
                t_idx = P.furthestsampling(xyz, offset, t_off)
                checks.append((t_idx != t_guess).any())
        else:
            with torch.cuda.stream(geo):
                t_idx = P.furthestsampling(xyz, offset, t_off)                            # :103, resumed from the stratified samples
                n_xyz = xyz[t_idx.long(), :].contiguous()
                made = torch.cuda.Event()
                made.record(geo)
            trans = grouping(n_xyz, made)
        trans["idx"] = t_idx
        STATS["transitions_prefetched"] += 1
        cloud.trans = trans
        _remember(trans["n_xyz"], _Cloud(trans["n_xyz"], t_host, True))
    seen = wrong = None
    if checks:
        STATS["speculated"] += 1
        with torch.cuda.stream(geo):
            wrong = _FLAG_HOST.get(dev.index)
            if wrong is None:  # (one pinned word per device, reused: a fresh pinned allocation per layer costs milliseconds now and then)
                wrong = _FLAG_HOST[dev.index] = [torch.empty(1, dtype=torch.bool, pin_memory=True) for _ in range(8)]
            wrong = wrong[STATS["speculated"] % 8]  # (a ring: the previous layers' words may not have been read yet)
            wrong.copy_(torch.stack(checks).any().reshape(1), non_blocking=True)
            seen = torch.cuda.Event()
            seen.record(geo)

    def body(samples):
        f = _blocks(self, feats, xyz, xyz, offset, samples, attn0, h, d, N)
        if down is not None:                                                              # :321-324
            return (f, xyz, offset) + tuple(down(f, xyz, offset))
        return f, xyz, offset, None, None, None

    if guess is None:
        main.wait_event(have_samples)
        sampled.record_stream(main)
        return body(sampled)
    out = body(guess)
    seen.synchronize()
    if not bool(wrong):
        return out
    # an exact tie among the samples: the identity prefix was not the sampler's answer - the stage once more with what the sampler
    # returned (and the TransitionDown geometry rebuilt from ITS samples); dropout / drop-path draw again
    STATS["reruns"] += 1
    del out
    main.wait_stream(geo)
    main.wait_stream(knn_s)
    if trans is not None:
        n_xyz = xyz[trans["idx"].long(), :].contiguous()
        knn_idx, _ = P.knnquery(int(down.k), xyz, n_xyz, offset, t_off)
        ready = torch.cuda.Event()
        ready.record(main)
        cloud.trans = dict(trans, n_xyz=n_xyz, knn=knn_idx, ready=ready)
        _remember(n_xyz, _Cloud(n_xyz, t_host, True))
    return body(sampled)


def transition_down_forward(self, feats, xyz, offset):
    """Replacement of TransitionDown.forward (:98-111), same arguments, same three results: sampling, the next cloud and the grouping
    query are taken from what the installed BasicLayer.forward put on the side streams (else the original forward runs); grouping,
    norm, linear and max-pool are the module's own."""
    cloud = _cloud_of(xyz)
    t = cloud.trans if cloud is not None else None
    if t is None or (t["ratio"], t["k"]) != (self.ratio, self.k) or not feats.is_cuda:
        return _original(self)(self, feats, xyz, offset)
    main = torch.cuda.current_stream(xyz.device)
    main.wait_event(t["ready"])
    for x in (t["n_xyz"], t["knn"], t["n_offset"]):
        x.record_stream(main)
    grouped = P.queryandgroup(int(self.k), xyz, t["n_xyz"], feats.contiguous(), t["knn"], offset, t["n_offset"], use_xyz=False)   # :104
    m, k, c = grouped.shape
    rows = grouped.view(m * k, c)
    if self.norm is not None:
        rows = self.norm(rows)
    pooled = self.pool(self.linear(rows.view(m, k, c)).transpose(1, 2).contiguous())                                                # :106-109
    return pooled.squeeze(-1), t["n_xyz"], t["n_offset"]


def patch_classes(basic_layer_cls=None, window_attention_cls=None, transition_down_cls=None):
    """Rebinds `forward` on the given classes (any classes with the reference's attribute names); returns what was patched."""
    done = []
    for cls, fn in ((basic_layer_cls, basic_layer_forward), (window_attention_cls, window_attention_forward),
                    (transition_down_cls, transition_down_forward)):
        if cls is None:
            continue
        if (cls, "forward") not in _ORIGINAL:
            _ORIGINAL[cls, "forward"] = cls.forward
        cls.forward = fn
        done.append(cls)
    return done


def install_fast_layers(module=None):
    """Patches BasicLayer / WindowAttention of the reference's model module(s).  module: the imported module (or a list);
    default: `model.stratified_transformer` (and `model.stratified_transformer_backup` when it is already imported), which must
    be importable - put the reference's root on sys.path and call stratified_transformer_amd.install() first."""
    import importlib
    if module is None:
        mods = [importlib.import_module("model.stratified_transformer")]
        if "model.stratified_transformer_backup" in sys.modules:
            mods.append(sys.modules["model.stratified_transformer_backup"])
    else:
        mods = list(module) if isinstance(module, (list, tuple)) else [module]
    done = []
    for m in mods:
        done += patch_classes(getattr(m, "BasicLayer", None), getattr(m, "WindowAttention", None), getattr(m, "TransitionDown", None))
    return done


def uninstall_fast_layers():
    for (cls, name), fn in list(_ORIGINAL.items()):
        setattr(cls, name, fn)
    _ORIGINAL.clear()
