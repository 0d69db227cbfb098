"""Optional fused fast path of WindowAttention.forward's operator sequence.

The reference model calls five operators per attention block
(model/stratified_transformer.py:183-208: attention_step1_v2, dot_prod_with_idx_v3, `+`, scatter_softmax,
attention_step2_with_rel_pos_value_v2).  `window_attention` below is ONE autograd function for the whole
sequence.  It is an addition, not a replacement: the model file runs unmodified on the operator API without
it; a caller that owns its attention module can call it instead and gets

  * forward: logits + softmax in one kernel (`window_logits_softmax_forward_launcher`: the key rows are
    gathered once instead of twice, the three [M, h] intermediates a1, a2, a1+a2 never exist), then the
    A4 kernel; only the softmax output [M, h] is kept for the backward;
  * backward: two walks over the pair list instead of seven (`window_attention_backward_launcher`: by query
    grad_attn -> softmax backward -> grad_logit and grad_q; by key grad_k and grad_v) plus the three table
    gradients; without a key-major view the operators' own backward launchers, in autograd's order;
  * the same numbers: every term is computed as the separate operators compute it (bit-identical for
    h = 3 or 4 heads, equal up to the order of the softmax sum otherwise).

d = 16 only (the shipped configs); other head dims: use the operators.
"""
import torch
from torch.autograd import Function

from . import _lib
from . import pointops as P
from . import pointops2_cuda as pointops_cuda
from ._lib import ptr


class WindowAttention(Function):
    @staticmethod
    def forward(ctx, q, k, v, table_q, table_k, table_v, index0_offsets, index1, rel_idx):
        for t in (q, k, v, table_q, table_k, table_v, index0_offsets, index1, rel_idx):
            assert t.is_contiguous()
        N, h, hdim = q.shape
        if hdim != 16:
            raise RuntimeError("window_attention: d != 16 (use the operators of pointops for other head dims)")
        M = index1.shape[0]
        L = table_q.shape[0]
        assert table_k.shape[0] == L and table_v.shape[0] == L
        P._check_pair_list("window_attention", N, index0_offsets, index1, rel_idx)
        pointops_cuda._chk((q, torch.float32, "q"), (k, torch.float32, "k"), (v, torch.float32, "v"),
                           (table_q, torch.float32, "table_q"), (table_k, torch.float32, "table_k"), (table_v, torch.float32, "table_v"),
                           (index0_offsets, torch.int32, "index0_offsets"), (index1, torch.int32, "index1"), (rel_idx, torch.int32, "rel_idx"))
        attn = torch.empty((M, h), dtype=torch.float32, device=q.device)
        pointops_cuda._rows(table_q)
        out = torch.zeros((N, h, hdim), dtype=torch.float32, device=q.device)
        with P._with_rows(P.row_order_of(index0_offsets, index1)):  # (the pair walkers take their rows in window order, csrc/common.h)
            pointops_cuda._call("window_logits_softmax_forward_launcher", q, int(index0_offsets.shape[0]) - 1, M, h, hdim,
                                ptr(q), ptr(index0_offsets), ptr(k), ptr(index1), ptr(table_q), ptr(table_k), ptr(rel_idx), ptr(attn))
            pointops_cuda.attention_step2_with_rel_pos_value_forward_cuda_v2(N, M, h, hdim, 0, attn, v, index0_offsets, index1, table_v, rel_idx, out)
        ctx.save_for_backward(q, k, v, table_q, table_k, table_v, index0_offsets, index1, rel_idx, attn)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, table_q, table_k, table_v, offs, index1, rel_idx, attn = ctx.saved_tensors
        N, h, hdim = q.shape
        NK = k.shape[0]
        M = index1.shape[0]
        L = table_q.shape[0]
        dev = q.device
        grad_out = grad_out.contiguous()
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
        e = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        csc = P.csc_of(offs, index1, NK)
        if csc is not None and L <= 80:
            # two walks (by query, by key) + the three table gradients: DESIGN.md 4.5
            grad_logit, grad_q, grad_k, grad_v = e(M, h), e(N, h, hdim), e(NK, h, hdim), e(v.shape[0], h, hdim)
            grad_tq, grad_tk, grad_tv = z(L, h, hdim, 3), z(L, h, hdim, 3), z(L, h, hdim, 3)
            pointops_cuda._chk((grad_out, torch.float32, "grad_out"))
            pointops_cuda._rows(table_q)
            with P._with_csc(csc), P._with_rows(P.row_order_of(offs, index1)):
                pointops_cuda._call("window_attention_backward_launcher", q, int(offs.shape[0]) - 1, M, h, hdim, ptr(grad_out), ptr(q), ptr(k),
                                    ptr(v), ptr(attn), ptr(offs), ptr(index1), ptr(table_q), ptr(table_k), ptr(table_v), ptr(rel_idx),
                                    ptr(grad_logit), ptr(grad_q), ptr(grad_k), ptr(grad_v), ptr(grad_tq), ptr(grad_tk), ptr(grad_tv))
            return grad_q, grad_k, grad_v, grad_tq, grad_tk, grad_tv, None, None, None
        with P._with_csc(csc), P._with_rows(P.row_order_of(offs, index1)):
            # the operators' own backward launchers, in autograd's order
            grad_attn = e(M, h)
            grad_v, grad_tv = z(v.shape[0], h, hdim), z(L, h, hdim, 3)
            pointops_cuda.attention_step2_with_rel_pos_value_backward_cuda_v2(N, M, h, hdim, 0, grad_out, offs, index1, attn, v, table_v, rel_idx,
                                                                              grad_attn, grad_v, grad_tv)
            grad_logit = e(M, h)
            pointops_cuda._call("segment_softmax_backward_launcher", attn, int(offs.shape[0]) - 1, M, h, ptr(attn), ptr(grad_attn), ptr(offs), ptr(grad_logit))
            # A1 and A2 receive the same gradient; grad_k is accumulated by both into one buffer
            gq1, gq2 = e(N, h, hdim), e(N, h, hdim)
            grad_k = z(NK, h, hdim)
            grad_tq, grad_tk = z(L, h, hdim, 3), z(L, h, hdim, 3)
            pointops_cuda.attention_step1_backward_cuda_v2(int(offs.shape[0]) - 1, M, h, h * hdim, 0, grad_logit, offs, index1, q, k, gq1, grad_k)
            pointops_cuda.dot_prod_with_idx_backward_cuda_v3(N, M, h, hdim, 0, grad_logit, q, offs, k, index1, table_q, table_k, rel_idx,
                                                             gq2, grad_k, grad_tq, grad_tk)
        return gq1.add_(gq2), grad_k, grad_v, grad_tq, grad_tk, grad_tv, None, None, None


def window_attention(q, k, v, table_q, table_k, table_v, index0_offsets, index1, rel_idx):
    """q, k, v [N, h, 16] f32 (q already scaled, as the model does at :181); tables [L, h, 16, 3]; the block's CSR
    pair list (index0_offsets [N+1], index1 [M] i32) and rel_idx [M, 3] i32  ->  [N, h, 16]"""
    return WindowAttention.apply(q, k, v, table_q, table_k, table_v, index0_offsets, index1, rel_idx)


class CellAttention(Function):
    """The same operator sequence on a cell plan (index_build.CellPlan; csrc/cell_attn.hip): one wave per
    (cell, head) loads the cell's key / value rows once for all its queries; the backward keeps dK / dV of a cell in
    registers and takes the three table gradients from cell-ordered softmax weights / logit gradients."""

    @staticmethod
    def forward(ctx, q, k, v, table_q, table_k, table_v, plan):
        N, h, hdim = q.shape
        if hdim != 16:
            raise RuntimeError("cell_attention: d != 16 (use the operators of pointops for other head dims)")
        L = table_q.shape[0]
        if plan.n_points != N or k.shape[0] != N or v.shape[0] != N:
            raise RuntimeError("cell_attention: the plan was built for %d points, q/k/v have %d/%d/%d rows" % (plan.n_points, N, k.shape[0], v.shape[0]))
        # The plan's packed rel-pos indices were clamped to [0, plan.table_rows) when it was filled and the kernels take the
        # tables' L as the axis stride of their LDS image: any other L would index another axis / table (the model asserts the
        # index range instead, model/stratified_transformer.py:189-190).
        if L != plan.table_rows:
            raise RuntimeError("cell_attention: the plan was built for tables of %d rows (cell_table_rows), the tables have %d" % (plan.table_rows, L))
        if L > 80 and any(ctx.needs_input_grad[:6]):
            raise RuntimeError("cell_attention: the backward supports at most 80 table rows (L = %d): use the operators of pointops, "
                               "or run the forward under torch.no_grad()" % L)
        st = q.dtype  # storage type of q / k / v / tables: fp32, or bf16 (fp32 arithmetic and outputs either way)
        if st not in (torch.float32, torch.bfloat16):
            raise TypeError("cell_attention: q / k / v / tables must all be float32 or all bfloat16, got %s" % st)
        pointops_cuda._chk((q, st, "q"), (k, st, "k"), (v, st, "v"), (table_q, st, "table_q"), (table_k, st, "table_k"), (table_v, st, "table_v"))
        assert table_k.shape == table_q.shape and table_v.shape == table_q.shape
        dev = q.device
        # a plan restricted to a share of the cells (CellPlan.share: one scene over several ranks) leaves the other cells' rows and
        # tile entries untouched: they must read as zero (the ranks' outputs are summed; the key-side table gradient walks all cells)
        alloc = torch.zeros if plan.partial else torch.empty
        out = alloc((N, h, hdim), dtype=torch.float32, device=dev)
        ml = torch.empty((N, h, 2), dtype=torch.float32, device=dev)
        pbuf = alloc((h, max(plan.n_pairs, 1)), dtype=torch.float32, device=dev)
        _lib.call("cell_attention_forward_launcher" if st == torch.float32 else "cell_attention_forward_bf16_launcher", plan.c_arg(), h, hdim, L, ptr(q), ptr(k), ptr(v), ptr(table_q), ptr(table_k), ptr(table_v),
                  ptr(out), ptr(ml), ptr(pbuf), device=dev)
        ctx.plan = plan
        ctx.save_for_backward(q, k, v, table_q, table_k, table_v, out, pbuf)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, table_q, table_k, table_v, out, pbuf = ctx.saved_tensors
        plan = ctx.plan
        N, h, hdim = q.shape
        L = table_q.shape[0]
        dev = q.device
        grad_out = grad_out.contiguous()
        pointops_cuda._chk((grad_out, torch.float32, "grad_out"))
        gsbuf = torch.zeros_like(pbuf) if plan.partial else torch.empty_like(pbuf)
        f32 = dict(dtype=torch.float32, device=dev)
        grad_q = (torch.zeros if plan.partial else torch.empty)(q.shape, **f32)
        # the five accumulated gradients are views of ONE zero-filled buffer: one fill kernel instead of five per block
        nkv, ntab = k.numel(), table_q.numel()
        acc = torch.zeros(2 * nkv + 3 * ntab, **f32)
        grad_k, grad_v = acc[:nkv].view(k.shape), acc[nkv:2 * nkv].view(v.shape)
        gtq, gtk, gtv = (acc[2 * nkv + i * ntab:2 * nkv + (i + 1) * ntab].view(table_q.shape) for i in range(3))
        _lib.call("cell_attention_backward_launcher" if q.dtype == torch.float32 else "cell_attention_backward_bf16_launcher", plan.c_arg(), h, hdim, L, ptr(grad_out), ptr(q), ptr(k), ptr(v), ptr(out), ptr(table_q),
                  ptr(table_k), ptr(table_v), ptr(pbuf), ptr(gsbuf), ptr(grad_q), ptr(grad_k), ptr(grad_v), ptr(gtq), ptr(gtk), ptr(gtv),
                  device=dev)
        if q.dtype != torch.float32:  # autograd wants a gradient of the operand's dtype; the sums above were taken in fp32
            grad_q, grad_k, grad_v, gtq, gtk, gtv = (g.to(q.dtype) for g in (grad_q, grad_k, grad_v, gtq, gtk, gtv))
        return grad_q, grad_k, grad_v, gtq, gtk, gtv, None


def cell_attention(q, k, v, table_q, table_k, table_v, plan):
    """q, k, v [N, h, 16] f32 (q already scaled, :181); tables [L, h, 16, 3]; plan = BlockIndex.cells of the block's
    pattern (index_build.stage_index_hip(..., cell_table_rows=L))  ->  [N, h, 16], same numbers as the five operators
    up to the order of the sums."""
    return CellAttention.apply(q, k, v, table_q, table_k, table_v, plan)
