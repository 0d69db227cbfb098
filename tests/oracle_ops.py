"""TEST INFRASTRUCTURE: the operator API of stratified_transformer_amd.pointops implemented on CPU tensors
with the oracle (oracle/pointops_ref.py), as torch.autograd.Functions.  Lets the distributed logic
(stratified_transformer_amd/sharding.py) run under gloo on CPU ranks.  Never imported by the product."""
import numpy as np
import torch
from torch.autograd import Function

from oracle import pointops_ref as ref


def _n(t):
    return t.detach().cpu().numpy()


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


class _A1(Function):
    @staticmethod
    def forward(ctx, q, k, index1, offsets, n_max):
        ctx.save_for_backward(q, k, index1, offsets)
        return _t(ref.attention_step1_v2(_n(q), _n(k), _n(index1), _n(offsets)))

    @staticmethod
    def backward(ctx, g):
        q, k, index1, offsets = ctx.saved_tensors
        gq, gk = ref.attention_step1_v2_backward(_n(g), _n(q), _n(k), _n(index1), _n(offsets))
        return _t(gq), _t(gk), None, None, None


class _A2(Function):
    @staticmethod
    def forward(ctx, q, offsets, n_max, k, index_k, tq, tk, rel):
        ctx.save_for_backward(q, offsets, k, index_k, tq, tk, rel)
        return _t(ref.dot_prod_with_idx_v3(_n(q), _n(offsets), _n(k), _n(index_k), _n(tq), _n(tk), _n(rel)))

    @staticmethod
    def backward(ctx, g):
        q, offsets, k, index_k, tq, tk, rel = ctx.saved_tensors
        gq, gk, gtq, gtk = ref.dot_prod_with_idx_v3_backward(_n(g), _n(q), _n(offsets), _n(k), _n(index_k), _n(tq), _n(tk), _n(rel))
        return _t(gq), None, None, _t(gk), None, _t(gtq), _t(gtk), None


class _A3(Function):
    @staticmethod
    def forward(ctx, src, offsets):
        y = _t(ref.segment_softmax(_n(src), _n(offsets)))
        ctx.save_for_backward(y, offsets)
        return y

    @staticmethod
    def backward(ctx, g):
        y, offsets = ctx.saved_tensors
        return _t(ref.segment_softmax_backward(_n(y), _n(g), _n(offsets))), None


class _A4(Function):
    @staticmethod
    def forward(ctx, attn, v, offsets, n_max, index1, table, rel):
        ctx.save_for_backward(attn, v, offsets, index1, table, rel)
        return _t(ref.attention_step2_with_rel_pos_value_v2(_n(attn), _n(v), _n(offsets), _n(index1), _n(table), _n(rel)))

    @staticmethod
    def backward(ctx, g):
        attn, v, offsets, index1, table, rel = ctx.saved_tensors
        ga, gv, gt = ref.attention_step2_with_rel_pos_value_v2_backward(_n(g), _n(attn), _n(v), _n(offsets), _n(index1), _n(table), _n(rel))
        return _t(ga), _t(gv), None, None, None, _t(gt), None


attention_step1_v2 = _A1.apply
dot_prod_with_idx_v3 = _A2.apply
segment_softmax = _A3.apply
attention_step2_with_rel_pos_value_v2 = _A4.apply
