"""Installable fast forms of the two model methods that drive the hot path, with the reference's own signatures:

    BasicLayer.forward(self, feats, xyz, offset)                                   model/stratified_transformer.py:267-326
    WindowAttention.forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max)            ...:164-217

`install_fast_layers()` (or `stratified_transformer_amd.install(fast_layers=True)`) rebinds these two methods on the classes of
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

Numbers: the same sums in another order (tests/test_hip_parity.py::test_installed_fast_layers_against_the_reference_layer:
output and every parameter gradient of a depth-2 BasicLayer with TransitionDown against the reference's own run, <= 1e-3).
"""
import sys

import torch

from . import fused, index_build
from . import pointops as P

_ORIGINAL = {}


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


def basic_layer_forward(self, feats, xyz, offset):
    """Replacement of BasicLayer.forward (:267-326), same arguments, same six results."""
    attn0 = self.blocks[0].attn
    if not (feats.is_cuda and attn0.rel_query and attn0.rel_key and attn0.rel_value):
        return _original(self)(self, feats, xyz, offset)
    h, d = _heads_dim(attn0)
    N = xyz.shape[0]
    xyz_c = xyz.float().contiguous()
    offset_i = offset.int().contiguous()
    off_host = offset_i.tolist()                                     # one host sync (the reference: one .item() per element, :283-287)
    P.hint_host_offsets(offset_i, off_host)
    new_host = index_build.stratified_new_offset(off_host, int(self.downsample_scale))
    new_offset = torch.tensor(new_host, dtype=torch.int32, device=xyz.device)
    P.hint_host_offsets(new_offset, new_host)
    downsample_idx = P.furthestsampling(xyz_c, offset_i, new_offset)                      # :289
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
    if self.downsample:                                                                   # :321-324
        feats_down, xyz_down, offset_down = self.downsample(feats, xyz, offset)
    else:
        feats_down, xyz_down, offset_down = None, None, None
    return feats, xyz, offset, feats_down, xyz_down, offset_down


def patch_classes(basic_layer_cls=None, window_attention_cls=None):
    """Rebinds `forward` on the given classes (any classes with the reference's attribute names); returns what was patched."""
    done = []
    for cls, fn in ((basic_layer_cls, basic_layer_forward), (window_attention_cls, window_attention_forward)):
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
        done += patch_classes(getattr(m, "BasicLayer", None), getattr(m, "WindowAttention", None))
    return done


def uninstall_fast_layers():
    for (cls, name), fn in list(_ORIGINAL.items()):
        setattr(cls, name, fn)
    _ORIGINAL.clear()
