"""Golden vectors of ONE WHOLE STAGE, produced by EXECUTING THE REFERENCE'S OWN `BasicLayer` on CPU (build container only; same
shims as make_golden.py, nothing of the reference is copied):

  model/stratified_transformer.py  BasicLayer.__init__/forward (:250-326: batch ids, grid_sample x4, the new_offset rule,
                                   get_indice_pairs + CSR for the even AND the odd block), SwinTransformerBlock (:219-248),
                                   WindowAttention (:114-217), Mlp (:66-83), TransitionDown (:87-111)
  lib/pointops2/functions/pointops.py  queryandgroup (:648-675, the reference's own torch code)

depth = 2 (one plain, one shifted block), channel 48 -> 96, 3 heads, window 0.16, quant 0.01 (L = 64), downsample_scale 8,
ratio 0.25, k = 16, two batch elements.  The compiled operators the reference would call are bound to:
  furthestsampling, knnquery        the CPU oracle (oracle/pointops_oracle.c - the reference has no Python form of them)
  attention_step1_v2, dot_prod_with_idx_v3, attention_step2_with_rel_pos_value_v2
                                    the pure-torch definitions of the reference's test scripts (make_golden.py)
torch.sort (:312) and torch.argsort (:63) are run stable (the canonical order).

    python tests/golden/make_golden_layer.py   ->  tests/golden/basic_layer_1400.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import make_golden as mg  # noqa: E402
from oracle import index_ref, pointops_ref as ref  # noqa: E402


def t_furthestsampling(xyz, offset, new_offset):
    return torch.from_numpy(ref.furthestsampling(xyz.detach().numpy(), offset.numpy().astype(np.int32), new_offset.numpy().astype(np.int32)))


def t_knnquery(nsample, xyz, new_xyz, offset, new_offset):
    idx, dist = ref.knnquery(nsample, xyz.detach().numpy(), (xyz if new_xyz is None else new_xyz).detach().numpy(),
                             offset.numpy().astype(np.int32), new_offset.numpy().astype(np.int32))
    return torch.from_numpy(idx), torch.from_numpy(dist)


def main():
    mg.install_shims()
    # `lib.pointops2.functions.pointops` must be the REFERENCE's module here (its queryandgroup is executed): this repository's
    # root holds a drop-in of the same dotted name, so it leaves sys.path before the model is imported
    root = os.path.realpath(os.path.join(HERE, "..", ".."))
    sys.path[:] = [p for p in sys.path if os.path.realpath(p or ".") != root]
    assert "lib" not in sys.modules
    sys.path.insert(0, mg.REF)
    import model.stratified_transformer as st  # the reference
    assert os.path.realpath(st.pointops.__file__).startswith(mg.REF), st.pointops.__file__
    st.pointops.attention_step1_v2 = mg.t_attention_step1_v2
    st.pointops.dot_prod_with_idx_v3 = mg.t_dot_prod_with_idx_v3
    st.pointops.attention_step2_with_rel_pos_value_v2 = mg.t_attention_step2_with_rel_pos_value_v2
    st.pointops.furthestsampling = t_furthestsampling
    st.pointops.knnquery = t_knnquery
    _sort = torch.sort
    torch.sort = lambda input, *a, **k: _sort(input, *a, **{**k, "stable": True}) if not a and "dim" not in k else _sort(input, *a, **k)

    C, C_out, h, w, quant, scale, depth = 48, 96, 3, 0.16, 0.01, 8, 2
    offset = torch.tensor([800, 1400], dtype=torch.int32)
    xyz = torch.cat([mg.synthetic_room(800, 21), mg.synthetic_room(600, 22, box=(0.7, 0.6, 0.5))], 0)
    # Coordinates on a 1 mm lattice.  torch divides by 100000 (:187) as a true division on CPU and as a multiplication by the
    # fp32 reciprocal on a GPU; the two disagree (by one bin) only for rounded offsets of exactly 0.0001 + j * quant, which a
    # lattice of 1 mm cannot produce - so this fixture is what the reference computes on EITHER device (asserted below),
    # and points sit exactly on window borders, where the fp32 division / floor-division rules decide.
    xyz = (torch.round(xyz * 1000) / 1000).contiguous()
    N = xyz.shape[0]
    ds_chk = t_furthestsampling(xyz, offset, torch.tensor(index_ref.stratified_new_offset(offset.numpy(), scale), dtype=torch.int32))
    for parity in (0, 1):
        a = index_ref.build_stage_indices(xyz, offset.numpy(), w, quant, ds_chk, parity, div_mode="cpu")
        b = index_ref.build_stage_indices(xyz, offset.numpy(), w, quant, ds_chk, parity, div_mode="cuda")
        assert torch.equal(a["rel_idx"], b["rel_idx"]), "rel-pos index depends on the device's arithmetic"
    torch.manual_seed(3)
    layer = st.BasicLayer(scale, depth, C, h, w, 0.04, quant, rel_query=True, rel_key=True, rel_value=True, drop_path=0.0,
                          downsample=st.TransitionDown, ratio=0.25, k=16, out_channels=C_out)
    with torch.no_grad():  # O(1) parameters so that every term matters
        for name, p in layer.named_parameters():
            if "relative_pos" in name:
                p.copy_(torch.randn(p.shape) * 0.5)
            elif name.endswith("weight") and p.dim() == 2:
                p.copy_(torch.randn(p.shape) * (1.0 / p.shape[1] ** 0.5))
            elif name.endswith("bias"):
                p.copy_(torch.randn(p.shape) * 0.1)
            elif name.endswith("weight") and p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn(p.shape))
    feats = torch.randn(N, C, requires_grad=True)
    f, x, o, f_down, x_down, o_down = layer(feats, xyz, offset)
    g = torch.Generator().manual_seed(5)
    g1, g2 = torch.randn(f.shape, generator=g), torch.randn(f_down.shape, generator=g)
    ((f * g1).sum() + (f_down * g2).sum()).backward()
    out = dict(xyz=xyz.numpy(), offset=offset.numpy(), feats=feats.detach().numpy(), grad_out=g1.numpy(), grad_out_down=g2.numpy(),
               out=f.detach().numpy(), out_down=f_down.detach().numpy(), xyz_down=x_down.numpy(), offset_down=o_down.numpy().astype(np.int32),
               grad_feats=feats.grad.numpy(), config=np.array([scale, depth, C, C_out, h, 16], dtype=np.int32),
               window_size=np.float64(w), quant_size=np.float64(quant))
    for name, p in layer.named_parameters():
        out["param." + name] = p.detach().numpy()
        out["grad." + name] = p.grad.numpy()
    path = os.path.join(HERE, "basic_layer_1400.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, round(os.path.getsize(path) / 1e6, 2), "MB;  N =", N, " m_down =", f_down.shape[0],
          " |out| =", float(f.abs().max()), " |grad_feats| =", float(feats.grad.abs().max()))


if __name__ == "__main__":
    main()
