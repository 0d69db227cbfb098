"""More golden vectors produced by EXECUTING THE REFERENCE'S OWN PYTHON on CPU (build container only; same recipe and
shims as make_golden.py, nothing of the reference is copied):

  window_attention_4000_h6.npz   model/stratified_transformer.py grid_sample (:44), get_indice_pairs (:10) and
                                 WindowAttention (:114-217) at BASELINE config-1 size: 4 000 points, C = 96, h = 6, window
                                 0.2, quant 0.01 -> L = 80 tables (stage 1 of the ScanNet yaml)
  swin3d_window_attention.npz    model/swin3d_transformer.py grid_sample (:11), the pair construction of BasicLayer.forward
                                 (:239-278, stable sort) and WindowAttention (:81-178: tables of 2*L-1 rows, rel-pos index
                                 = quantised in-window coordinates' difference + L - 1, :149-154) for the plain and the
                                 shifted pattern

    python tests/golden/make_golden_extra.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402
from oracle import index_ref  # noqa: E402


def t_dot_prod_with_idx(q, index, table, rel_idx):  # test_relative_pos_encoding_op_step1.py:26-30
    return (q[index.long()] * mg._rel_enc(table, rel_idx)).sum(-1)


def t_attention_step2(attn, v, index0, index1):  # test_attention_op_step2.py:25-29
    out = attn.unsqueeze(-1) * v[index1.long()]
    return torch.zeros(v.shape).index_add_(0, index0.long(), out)


def small16(a):
    a = np.asarray(a)
    return a.astype(np.int16) if a.size and a.min() >= -32768 and a.max() < 32768 else a.astype(np.int32)


def stratified_h6(st):
    out = {}
    N, C, h, w, quant, scale = 4000, 96, 6, 0.2, 0.01, 4
    offset = np.array([N], dtype=np.int32)
    xyz = mg.synthetic_room(N, 3, box=(1.6, 1.3, 0.9))
    batch = index_ref.batch_from_offset(offset)
    g = torch.Generator().manual_seed(11)
    new_offset = index_ref.stratified_new_offset(offset, scale)
    downsample_idx = torch.randperm(N, generator=g)[: int(new_offset[0])].sort()[0].int()
    ws = torch.tensor([w] * 3).type_as(xyz)
    small = st.grid_sample(xyz, batch, ws, start=None)
    large = st.grid_sample(xyz, batch, 2 * ws, start=None)
    i0, i1 = st.get_indice_pairs(small[1], small[2], large[1], large[2], downsample_idx, batch, xyz, ws, 0)
    i0, indices = torch.sort(i0, stable=True)  # :312-317, stable
    i1 = i1[indices]
    cnts = i0.bincount()
    n_max = cnts.max()
    offs = torch.cat([torch.zeros(1, dtype=torch.long), cnts.cumsum(dim=-1)], 0)
    rel = xyz[i0] - xyz[i1]
    rel = torch.round(rel * 100000) / 100000
    rel_idx = (rel + 2 * w - 0.0001) // quant
    torch.manual_seed(1)
    attn = st.WindowAttention(C, w, h, quant, rel_query=True, rel_key=True, rel_value=True)
    assert attn.relative_pos_query_table.shape[0] == 80
    with torch.no_grad():
        for p in (attn.relative_pos_query_table, attn.relative_pos_key_table, attn.relative_pos_value_table):
            p.copy_(torch.randn(p.shape) * 0.5)
        attn.qkv.weight.copy_(torch.randn(attn.qkv.weight.shape) * 0.1)
        attn.qkv.bias.copy_(torch.randn(attn.qkv.bias.shape) * 0.1)
        attn.proj.weight.copy_(torch.randn(attn.proj.weight.shape) * 0.1)
    feats = torch.randn(N, C, requires_grad=True)
    y = attn(feats, xyz, i0, i1, offs, n_max)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    out.update(xyz=xyz.numpy(), offset=offset, downsample_idx=small16(downsample_idx.numpy()), window_size=np.float64(w), quant_size=np.float64(quant),
               index_0=small16(i0.numpy()), index_1=small16(i1.numpy()), offsets=offs.numpy().astype(np.int32), n_max=np.int32(int(n_max)),
               rel_idx_cpu=rel_idx.numpy().astype(np.int8),
               feats=feats.detach().numpy(), grad_out=gy.numpy(), out=y.detach().numpy(), grad_feats=feats.grad.numpy(),
               qkv_weight=attn.qkv.weight.detach().numpy(), qkv_bias=attn.qkv.bias.detach().numpy(),
               proj_weight=attn.proj.weight.detach().numpy(), proj_bias=attn.proj.bias.detach().numpy(),
               table_q=attn.relative_pos_query_table.detach().numpy(), table_k=attn.relative_pos_key_table.detach().numpy(),
               table_v=attn.relative_pos_value_table.detach().numpy(),
               grad_table_q=attn.relative_pos_query_table.grad.numpy(), grad_table_k=attn.relative_pos_key_table.grad.numpy(),
               grad_table_v=attn.relative_pos_value_table.grad.numpy(), grad_qkv_weight=attn.qkv.weight.grad.numpy())
    path = os.path.join(HERE, "window_attention_4000_h6.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB;  M =", int(i0.shape[0]), " n_max =", int(n_max))


def swin3d(sw):
    out = {}
    N, C, h, w, quant = 2000, 48, 3, 0.16, 0.01   # quant_grid_length 16 -> tables of 31 rows (the reference's own test size, L = 31)
    offset = np.array([1200, 2000], dtype=np.int32)
    xyz = torch.cat([mg.synthetic_room(1200, 5), mg.synthetic_room(800, 6, box=(0.7, 0.6, 0.5))], 0)
    batch = index_ref.batch_from_offset(offset)
    g = torch.Generator().manual_seed(13)
    ws = torch.tensor([w] * 3).type_as(xyz)
    out.update(xyz=xyz.numpy(), offset=offset, window_size=np.float64(w), quant_size=np.float64(quant))
    torch.manual_seed(2)
    attn = sw.WindowAttention(C, w, h, quant, rel_query=True, rel_key=True, rel_value=True)
    assert attn.relative_pos_query_table.shape[0] == 31
    with torch.no_grad():
        for p in (attn.relative_pos_query_table, attn.relative_pos_key_table, attn.relative_pos_value_table):
            p.copy_(torch.randn(p.shape) * 0.5)
        attn.qkv.weight.copy_(torch.randn(attn.qkv.weight.shape) * 0.15)
        attn.qkv.bias.copy_(torch.randn(attn.qkv.bias.shape) * 0.1)
        attn.proj.weight.copy_(torch.randn(attn.proj.weight.shape) * 0.15)
    out.update(qkv_weight=attn.qkv.weight.detach().numpy(), qkv_bias=attn.qkv.bias.detach().numpy(), proj_weight=attn.proj.weight.detach().numpy(),
               proj_bias=attn.proj.bias.detach().numpy(), table_q=attn.relative_pos_query_table.detach().numpy(),
               table_k=attn.relative_pos_key_table.detach().numpy(), table_v=attn.relative_pos_value_table.detach().numpy())
    feats0 = torch.randn(N, C)
    out["feats"] = feats0.numpy()
    for pat, (pos, start, shift) in enumerate(((xyz, None, 0.0), (xyz + 1 / 2 * ws, xyz.min(0)[0], 1 / 2 * ws))):
        v2p, p2v, counts = sw.grid_sample(pos, batch, ws, start=start)
        n, k = p2v.shape
        mask = torch.arange(k).unsqueeze(0) < counts.unsqueeze(-1)  # swin3d_transformer.py:245-250
        mask_mat = mask.unsqueeze(-1) & mask.unsqueeze(-2)
        i0 = p2v.unsqueeze(-1).expand(-1, -1, k)[mask_mat]
        i1 = p2v.unsqueeze(1).expand(-1, k, -1)[mask_mat]
        i0, indices = torch.sort(i0, stable=True)  # :252-258, stable
        i1 = i1[indices]
        cnts = i0.bincount()
        n_max = cnts.max()
        offs = torch.cat([torch.zeros(1, dtype=torch.long), cnts.cumsum(dim=-1)], 0)
        # :151-154 evaluated by torch CPU
        xyz_quant = (xyz - xyz.min(0)[0] + shift) % w
        xyz_quant = xyz_quant // quant
        rel_idx = (xyz_quant[i0] - xyz_quant[i1]) + attn.quant_grid_length - 1
        for p in attn.parameters():
            p.grad = None
        feats = feats0.clone().requires_grad_(True)
        y = attn(feats, xyz, i0, offs, n_max, i1, shift)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        out.update({f"p{pat}_index_0": small16(i0.numpy()), f"p{pat}_index_1": small16(i1.numpy()), f"p{pat}_offsets": offs.numpy().astype(np.int32),
                    f"p{pat}_n_max": np.int32(int(n_max)), f"p{pat}_rel_idx_cpu": rel_idx.numpy().astype(np.int8),
                    f"p{pat}_out": y.detach().numpy(), f"p{pat}_grad_out": gy.numpy(), f"p{pat}_grad_feats": feats.grad.numpy(),
                    f"p{pat}_grad_table_q": attn.relative_pos_query_table.grad.numpy().copy(),
                    f"p{pat}_grad_table_k": attn.relative_pos_key_table.grad.numpy().copy(),
                    f"p{pat}_grad_table_v": attn.relative_pos_value_table.grad.numpy().copy()})
        print("swin3d pattern", pat, "M =", int(i0.shape[0]), "n_max =", int(n_max), "rel range", int(rel_idx.min()), int(rel_idx.max()))
    path = os.path.join(HERE, "swin3d_window_attention.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB")


def main():
    mg.install_shims()
    sys.path.insert(0, mg.REF)
    import model.stratified_transformer as st  # the reference
    import model.swin3d_transformer as sw      # the reference
    for mod in (st, sw):
        mod.pointops.attention_step1_v2 = mg.t_attention_step1_v2
        mod.pointops.dot_prod_with_idx_v3 = mg.t_dot_prod_with_idx_v3
        mod.pointops.attention_step2_with_rel_pos_value_v2 = mg.t_attention_step2_with_rel_pos_value_v2
        mod.pointops.dot_prod_with_idx = t_dot_prod_with_idx
        mod.pointops.attention_step2 = t_attention_step2
    stratified_h6(st)
    swin3d(sw)


if __name__ == "__main__":
    main()
