"""Generates tests/golden/*.npz by EXECUTING THE REFERENCE'S OWN PYTHON on CPU.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py

What is executed from the reference (imported from /root/reference, nothing is copied):
  model/stratified_transformer.py: grid_sample (:44), get_indice_pairs (:10),
  WindowAttention.__init__/forward (:114-217).
The imports the reference needs but this image lacks are shimmed with stand-ins that carry no
reference code (recipe: SURVEY.md §8c): torch_points3d / timm (stubs, off the hot path),
torch_scatter.scatter_softmax and torch_geometric.nn.voxel_grid (third-party restatements in
oracle/index_ref.py — parity unpinned at that boundary), an empty `pointops2_cuda`, and
`Tensor.cuda()` as identity.  The three pointops ops called by WindowAttention.forward are bound
to the pure-torch definitions the reference's own test scripts give
(lib/pointops2/functions/test_attention_op_step2.py:25-29,
 test_relative_pos_encoding_op_step1.py:26-30, test_relative_pos_encoding_op_step2.py:32-38).

The CSR step (stratified_transformer.py:312-317) is inline in BasicLayer.forward, so it is
re-stated here with stable=True (the canonical order).  The FPS subset is an input
of the fixture (FPS itself has no Python form in the reference).
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

from oracle import index_ref  # noqa: E402


def install_shims():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Stub(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    mod("torch_points3d")
    mod("torch_points3d.modules")
    mod("torch_points3d.modules.KPConv")
    mod("torch_points3d.modules.KPConv.kernels", KPConvLayer=_Stub)
    mod("torch_points3d.core")
    mod("torch_points3d.core.common_modules", FastBatchNorm1d=_Stub)
    mod("timm")
    mod("timm.models")
    mod("timm.models.layers", DropPath=_Stub, trunc_normal_=torch.nn.init.trunc_normal_)
    mod("torch_scatter", scatter_softmax=lambda src, index, dim=0: index_ref.scatter_softmax(src, index))
    mod("torch_geometric")
    mod("torch_geometric.nn", voxel_grid=index_ref.voxel_grid)
    mod("pointops2_cuda")
    torch.Tensor.cuda = lambda self, *a, **k: self
    # grid_sample (:63) calls torch.argsort without stable=True, so the order of points inside a
    # window is unspecified in the reference (and differs run to run on a GPU).  Pin the canonical
    # order: make that call stable.  Every order is a valid reference output.
    _argsort = torch.argsort
    torch.argsort = lambda input, *a, **k: _argsort(input, *a, **{**k, "stable": True})
    torch.cuda.IntTensor = torch.IntTensor
    torch.cuda.FloatTensor = torch.FloatTensor


# ---- pure-torch op definitions, after the reference's test scripts ---------------------------
def t_attention_step1_v2(q, k, index1, index0_offsets, n_max):
    counts = (index0_offsets[1:] - index0_offsets[:-1]).long()
    index0 = torch.repeat_interleave(torch.arange(counts.shape[0]), counts)
    return (q[index0] * k[index1.long()]).sum(-1)  # test_attention_op_step1.py:26-37 semantics


def _rel_enc(table, rel_idx):
    rel_idx = rel_idx.long()
    return table[:, :, :, 0][rel_idx[:, 0]] + table[:, :, :, 1][rel_idx[:, 1]] + table[:, :, :, 2][rel_idx[:, 2]]


def t_dot_prod_with_idx_v3(q, index_q_offsets, n_max, k, index_k, table_q, table_k, rel_idx):
    counts = (index_q_offsets[1:] - index_q_offsets[:-1]).long()
    index_q = torch.repeat_interleave(torch.arange(counts.shape[0]), counts)
    # test_relative_pos_encoding_op_step1.py:26-30, applied to (q, table_q) and (k, table_k) and
    # summed as test_relative_pos_encoding_op_step1_v3.py:60-62 does
    return (q[index_q] * _rel_enc(table_q, rel_idx)).sum(-1) + (k[index_k.long()] * _rel_enc(table_k, rel_idx)).sum(-1)


def t_attention_step2_with_rel_pos_value_v2(attn, v, index0_offsets, n_max, index1, table, rel_idx):
    counts = (index0_offsets[1:] - index0_offsets[:-1]).long()
    index0 = torch.repeat_interleave(torch.arange(counts.shape[0]), counts)
    out = attn.unsqueeze(-1) * (v[index1.long()] + _rel_enc(table, rel_idx))  # ...op_step2.py:32-37
    return torch.zeros(v.shape[0], v.shape[1], v.shape[2]).index_add_(0, index0, out)  # scatter_sum :38


def synthetic_room(n, seed, box=(0.8, 0.8, 0.6)):
    """Floor + two wall planes of a small box at S3DIS surface density (~625 pts/m^2 after 0.04 m
    voxelisation), seeded: gives M/N ~ 45 and a stratified share like BASELINE config 2."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n, 3, generator=g)
    which = torch.randint(0, 3, (n,), generator=g)
    xyz = u * torch.tensor(box)
    xyz[which == 0, 2] = 0.0
    xyz[which == 1, 0] = 0.0
    xyz[which == 2, 1] = 0.0
    xyz = xyz + 0.002 * torch.randn(n, 3, generator=g)
    xyz = xyz - xyz.min(0)[0]
    return xyz.float().contiguous()


def main():
    install_shims()
    sys.path.insert(0, REF)
    import model.stratified_transformer as st  # the reference

    st.pointops.attention_step1_v2 = t_attention_step1_v2
    st.pointops.dot_prod_with_idx_v3 = t_dot_prod_with_idx_v3
    st.pointops.attention_step2_with_rel_pos_value_v2 = t_attention_step2_with_rel_pos_value_v2

    out = {}
    N, C, h, w, quant, scale = 1000, 48, 3, 0.16, 0.01, 8
    offset = np.array([600, 1000], dtype=np.int32)  # two batch elements
    xyz = torch.cat([synthetic_room(600, 0), synthetic_room(400, 1, box=(0.7, 0.6, 0.5))], 0)
    batch = index_ref.batch_from_offset(offset)
    g = torch.Generator().manual_seed(7)
    # stratified key subset: an input of the fixture (random per batch element, sizes as :283-288)
    new_offset = index_ref.stratified_new_offset(offset, scale)
    ds = []
    lo, mlo = 0, 0
    for b in range(len(offset)):
        m_b = int(new_offset[b]) - mlo
        ds.append(lo + torch.randperm(int(offset[b]) - lo, generator=g)[:m_b].sort()[0])
        lo, mlo = int(offset[b]), int(new_offset[b])
    downsample_idx = torch.cat(ds).int()
    out.update(xyz=xyz.numpy(), offset=offset, new_offset=new_offset, downsample_idx=downsample_idx.numpy(),
               window_size=np.float64(w), quant_size=np.float64(quant))

    ws = torch.tensor([w] * 3).type_as(xyz)
    grids = {
        "small": st.grid_sample(xyz, batch, ws, start=None),
        "small_shift": st.grid_sample(xyz + 1 / 2 * ws, batch, ws, start=xyz.min(0)[0]),
        "large": st.grid_sample(xyz, batch, 2 * ws, start=None),
        "large_shift": st.grid_sample(xyz + 1 / 2 * (2 * ws), batch, 2 * ws, start=xyz.min(0)[0]),
    }
    for name, (cl, p2v, cnt) in grids.items():
        out[f"grid_{name}_cluster"] = cl.numpy().astype(np.int32)
        out[f"grid_{name}_p2v"] = p2v.numpy().astype(np.int32)
        out[f"grid_{name}_counts"] = cnt.numpy().astype(np.int32)

    blocks = {}
    for i in (0, 1):
        s, l = ("small", "large") if i == 0 else ("small_shift", "large_shift")
        i0, i1 = st.get_indice_pairs(grids[s][1], grids[s][2], grids[l][1], grids[l][2], downsample_idx, batch, xyz, ws, i)
        out[f"blk{i}_pairs_unsorted_index_0"] = i0.numpy().astype(np.int32)
        out[f"blk{i}_pairs_unsorted_index_1"] = i1.numpy().astype(np.int32)
        # stratified_transformer.py:312-317, stable
        i0, indices = torch.sort(i0, stable=True)
        i1 = i1[indices]
        cnts = i0.bincount()
        n_max = cnts.max()
        offs = torch.cat([torch.zeros(1, dtype=torch.long), cnts.cumsum(dim=-1)], 0)
        blocks[i] = (i0, i1, offs, n_max)
        out[f"blk{i}_index_0"] = i0.numpy().astype(np.int32)
        out[f"blk{i}_index_1"] = i1.numpy().astype(np.int32)
        out[f"blk{i}_offsets"] = offs.numpy().astype(np.int32)
        out[f"blk{i}_n_max"] = np.int32(int(n_max))
        # stratified_transformer.py:186-188 evaluated by torch CPU (true division by 100000)
        rel = xyz[i0] - xyz[i1]
        rel = torch.round(rel * 100000) / 100000
        rel_idx = (rel + 2 * w - 0.0001) // quant
        out[f"blk{i}_rel_idx_cpu"] = rel_idx.int().numpy()

    # ---- WindowAttention forward/backward through the reference module (block 0 pattern) -----
    torch.manual_seed(0)
    attn = st.WindowAttention(C, w, h, quant, rel_query=True, rel_key=True, rel_value=True)
    with torch.no_grad():  # make tables and biases O(1) so every term matters
        for p in (attn.relative_pos_query_table, attn.relative_pos_key_table, attn.relative_pos_value_table):
            p.copy_(torch.randn(p.shape) * 0.5)
        attn.qkv.weight.copy_(torch.randn(attn.qkv.weight.shape) * 0.15)
        attn.qkv.bias.copy_(torch.randn(attn.qkv.bias.shape) * 0.1)
        attn.proj.weight.copy_(torch.randn(attn.proj.weight.shape) * 0.15)
    feats = torch.randn(N, C, requires_grad=True)
    i0, i1, offs, n_max = blocks[0]

    rec = {}

    def tap(name, fn):
        def wrapped(*a):
            y = fn(*a)
            y.retain_grad()
            rec[name] = (a, y)
            return y
        return wrapped

    st.pointops.attention_step1_v2 = tap("a1", t_attention_step1_v2)
    st.pointops.dot_prod_with_idx_v3 = tap("a2", t_dot_prod_with_idx_v3)
    st.pointops.attention_step2_with_rel_pos_value_v2 = tap("a4", t_attention_step2_with_rel_pos_value_v2)
    real_ss = st.scatter_softmax

    def ss(src, index, dim=0):
        src.retain_grad()
        y = real_ss(src=src, index=index, dim=dim)
        y.retain_grad()
        rec["a3"] = ((src, index), y)
        return y

    st.scatter_softmax = ss
    y = attn(feats, xyz, i0, i1, offs, n_max)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)

    q, k = rec["a1"][0][0], rec["a1"][0][1]
    v = rec["a4"][0][1]
    # q/k/v are non-leaf: recover their grads through qkv = Linear(feats); store the module-level
    # results and the op-level tensors separately.
    out.update(
        wa_feats=feats.detach().numpy(), wa_grad_out=gy.numpy(), wa_out=y.detach().numpy(),
        wa_grad_feats=feats.grad.numpy(),
        wa_qkv_weight=attn.qkv.weight.detach().numpy(), wa_qkv_bias=attn.qkv.bias.detach().numpy(),
        wa_proj_weight=attn.proj.weight.detach().numpy(), wa_proj_bias=attn.proj.bias.detach().numpy(),
        wa_table_q=attn.relative_pos_query_table.detach().numpy(), wa_table_k=attn.relative_pos_key_table.detach().numpy(),
        wa_table_v=attn.relative_pos_value_table.detach().numpy(),
        wa_grad_table_q=attn.relative_pos_query_table.grad.numpy(), wa_grad_table_k=attn.relative_pos_key_table.grad.numpy(),
        wa_grad_table_v=attn.relative_pos_value_table.grad.numpy(),
        wa_grad_qkv_weight=attn.qkv.weight.grad.numpy(),
        op_q=q.detach().numpy(), op_k=k.detach().numpy(), op_v=v.detach().numpy(),
        # a3_in = a1_out + a2_out; grad(a1_out) = grad(a2_out) = a3_grad_in; grad_attn(a4) = a3_grad_out
        op_a1_out=rec["a1"][1].detach().numpy(), op_a2_out=rec["a2"][1].detach().numpy(),
        op_a3_out=rec["a3"][1].detach().numpy(),
        op_a3_grad_out=rec["a3"][1].grad.numpy(), op_a3_grad_in=rec["a3"][0][0].grad.numpy(),
        op_a4_out=rec["a4"][1].detach().numpy(), op_a4_grad_out=rec["a4"][1].grad.numpy(),
    )

    # ---- op-level input grads: each pure-torch definition alone, with the grad_out recorded above
    def leaf(t):
        return t.detach().clone().requires_grad_(True)

    tq, tk, tv = (leaf(attn.relative_pos_query_table), leaf(attn.relative_pos_key_table), leaf(attn.relative_pos_value_table))
    rel_idx = torch.from_numpy(out["blk0_rel_idx_cpu"])
    ql, kl = leaf(q), leaf(k)
    t_attention_step1_v2(ql, kl, i1.int(), offs.int(), n_max).backward(rec["a1"][1].grad)
    out.update(op_a1_grad_q=ql.grad.numpy(), op_a1_grad_k=kl.grad.numpy())
    ql, kl = leaf(q), leaf(k)
    t_dot_prod_with_idx_v3(ql, offs.int(), n_max, kl, i1.int(), tq, tk, rel_idx).backward(rec["a2"][1].grad)
    out.update(op_a2_grad_q=ql.grad.numpy(), op_a2_grad_k=kl.grad.numpy(), op_a2_grad_table_q=tq.grad.numpy(), op_a2_grad_table_k=tk.grad.numpy())
    al, vl = leaf(rec["a3"][1]), leaf(v)
    t_attention_step2_with_rel_pos_value_v2(al, vl, offs.int(), n_max, i1.int(), tv, rel_idx).backward(rec["a4"][1].grad)
    assert torch.allclose(al.grad, rec["a3"][1].grad, rtol=1e-5, atol=1e-6)
    out.update(op_a4_grad_v=vl.grad.numpy(), op_a4_grad_table=tv.grad.numpy())

    path = os.path.join(HERE, "window_attention_1000.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB;  M =", int(i0.shape[0]), " n_max =", int(n_max))
    for k_, v_ in sorted(out.items()):
        print(f"  {k_:32s} {getattr(v_, 'shape', ())} {getattr(v_, 'dtype', type(v_))}")


if __name__ == "__main__":
    main()
