"""CPU: pins the oracle (oracle/) against the golden vectors the reference itself produced.

Bar: bit-exact for every integer tensor; fp32 ops within 2e-5 absolute (the golden side is torch's
own summation order, the oracle follows the CUDA kernels' order; the reference's own acceptance is
max squared error < 1e-8, lib/pointops2/functions/test_attention_op_step1.py:74).
"""
import numpy as np
import torch

from oracle import index_ref, pointops_ref as ref

TOL = dict(rtol=2e-5, atol=2e-5)


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _grids(g):
    xyz = _t(g["xyz"])
    batch = index_ref.batch_from_offset(g["offset"])
    w = float(g["window_size"])
    ws = torch.tensor([w] * 3).type_as(xyz)
    return xyz, batch, w, {
        "small": index_ref.grid_sample(xyz, batch, ws, None),
        "small_shift": index_ref.grid_sample(xyz + 1 / 2 * ws, batch, ws, xyz.min(0)[0]),
        "large": index_ref.grid_sample(xyz, batch, 2 * ws, None),
        "large_shift": index_ref.grid_sample(xyz + 1 / 2 * (2 * ws), batch, 2 * ws, xyz.min(0)[0]),
    }


def test_grid_sample_matches_reference(golden):
    _, _, _, grids = _grids(golden)
    for name, (cl, p2v, cnt) in grids.items():
        assert np.array_equal(cl.numpy(), golden[f"grid_{name}_cluster"]), name
        assert np.array_equal(p2v.numpy(), golden[f"grid_{name}_p2v"]), name
        assert np.array_equal(cnt.numpy(), golden[f"grid_{name}_counts"]), name


def test_pairs_csr_and_rel_idx_match_reference(golden):
    xyz, _, w, grids = _grids(golden)
    ds = _t(golden["downsample_idx"])
    for i in (0, 1):
        s, l = ("small", "large") if i == 0 else ("small_shift", "large_shift")
        i0, i1 = index_ref.get_indice_pairs(grids[s][1], grids[s][2], grids[l][1], grids[l][2], ds, xyz, w, i)
        assert np.array_equal(i0.numpy(), golden[f"blk{i}_pairs_unsorted_index_0"])
        assert np.array_equal(i1.numpy(), golden[f"blk{i}_pairs_unsorted_index_1"])
        i0, i1, offs, n_max = index_ref.csr_from_pairs(i0, i1, xyz.shape[0])
        assert np.array_equal(i0.numpy(), golden[f"blk{i}_index_0"])
        assert np.array_equal(i1.numpy(), golden[f"blk{i}_index_1"])
        assert np.array_equal(offs.numpy(), golden[f"blk{i}_offsets"])
        assert n_max == int(golden[f"blk{i}_n_max"])
        rel = index_ref.rel_pos_index(xyz, i0, i1, w, float(golden["quant_size"]), div_mode="cpu")
        assert np.array_equal(rel.numpy(), golden[f"blk{i}_rel_idx_cpu"])
        # the GPU ("cuda") arithmetic may only differ where round(x*1e5)/1e5 != round(x*1e5)*(1/1e5) flips a floor
        rel_gpu = index_ref.rel_pos_index(xyz, i0, i1, w, float(golden["quant_size"]), div_mode="cuda")
        assert (rel_gpu.numpy() != golden[f"blk{i}_rel_idx_cpu"]).mean() < 1e-3
        assert np.abs(rel_gpu.numpy() - golden[f"blk{i}_rel_idx_cpu"]).max() <= 1


def test_build_stage_indices_is_the_same_pipeline(golden):
    xyz = _t(golden["xyz"])
    for i in (0, 1):
        r = index_ref.build_stage_indices(xyz, golden["offset"], float(golden["window_size"]), float(golden["quant_size"]),
                                          _t(golden["downsample_idx"]), i, div_mode="cpu")
        assert np.array_equal(r["index_1"].numpy(), golden[f"blk{i}_index_1"])
        assert np.array_equal(r["offsets"].numpy(), golden[f"blk{i}_offsets"])
        assert np.array_equal(r["rel_idx"].numpy(), golden[f"blk{i}_rel_idx_cpu"])


def test_offset_rules():
    # stratified_transformer.py:283-288 and :98-102 evaluated by hand
    assert index_ref.stratified_new_offset([600, 1000], 8).tolist() == [76, 127]
    assert index_ref.transition_down_offset([600, 1000], 0.25).tolist() == [151, 252]
    # float accumulation for b>0: 10*.25+1 = 3.5 per element after the first
    assert index_ref.transition_down_offset([10, 20, 30], 0.25).tolist() == [3, 6, 10]


def test_attention_step1_v2(golden):
    g = golden
    out = ref.attention_step1_v2(g["op_q"], g["op_k"], g["blk0_index_1"], g["blk0_offsets"])
    np.testing.assert_allclose(out, g["op_a1_out"], **TOL)
    gq, gk = ref.attention_step1_v2_backward(g["op_a3_grad_in"], g["op_q"], g["op_k"], g["blk0_index_1"], g["blk0_offsets"])
    np.testing.assert_allclose(gq, g["op_a1_grad_q"], **TOL)
    np.testing.assert_allclose(gk, g["op_a1_grad_k"], **TOL)
    # v1 (pair-indexed) form computes the same thing
    out1 = ref.attention_step1(g["op_q"], g["op_k"], g["blk0_index_0"], g["blk0_index_1"])
    np.testing.assert_allclose(out1, g["op_a1_out"], **TOL)
    gq1, gk1 = ref.attention_step1_backward(g["op_a3_grad_in"], g["op_q"], g["op_k"], g["blk0_index_0"], g["blk0_index_1"])
    np.testing.assert_allclose(gq1, g["op_a1_grad_q"], **TOL)
    np.testing.assert_allclose(gk1, g["op_a1_grad_k"], **TOL)


def test_dot_prod_with_idx_v3(golden):
    g = golden
    args = (g["op_q"], g["blk0_offsets"], g["op_k"], g["blk0_index_1"], g["wa_table_q"], g["wa_table_k"], g["blk0_rel_idx_cpu"])
    np.testing.assert_allclose(ref.dot_prod_with_idx_v3(*args), g["op_a2_out"], **TOL)
    gq, gk, gtq, gtk = ref.dot_prod_with_idx_v3_backward(g["op_a3_grad_in"], *args)
    np.testing.assert_allclose(gq, g["op_a2_grad_q"], **TOL)
    np.testing.assert_allclose(gk, g["op_a2_grad_k"], **TOL)
    np.testing.assert_allclose(gtq, g["op_a2_grad_table_q"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(gtk, g["op_a2_grad_table_k"], rtol=1e-4, atol=1e-4)
    # v1 single-table form: sum of two calls == v3 (test_relative_pos_encoding_op_step1_v3.py:60-62)
    v1 = ref.dot_prod_with_idx(g["op_q"], g["blk0_index_0"], g["wa_table_q"], g["blk0_rel_idx_cpu"]) + \
        ref.dot_prod_with_idx(g["op_k"], g["blk0_index_1"], g["wa_table_k"], g["blk0_rel_idx_cpu"])
    np.testing.assert_allclose(v1, g["op_a2_out"], **TOL)


def test_segment_softmax(golden):
    g = golden
    y = ref.segment_softmax(g["op_a1_out"] + g["op_a2_out"], g["blk0_offsets"])
    np.testing.assert_allclose(y, g["op_a3_out"], **TOL)
    gx = ref.segment_softmax_backward(g["op_a3_out"], g["op_a3_grad_out"], g["blk0_offsets"])
    np.testing.assert_allclose(gx, g["op_a3_grad_in"], **TOL)


def test_attention_step2_with_rel_pos_value_v2(golden):
    g = golden
    args = (g["op_a3_out"], g["op_v"], g["blk0_offsets"], g["blk0_index_1"], g["wa_table_v"], g["blk0_rel_idx_cpu"])
    np.testing.assert_allclose(ref.attention_step2_with_rel_pos_value_v2(*args), g["op_a4_out"], **TOL)
    ga, gv, gt = ref.attention_step2_with_rel_pos_value_v2_backward(g["op_a4_grad_out"], *args)
    np.testing.assert_allclose(ga, g["op_a3_grad_out"], **TOL)
    np.testing.assert_allclose(gv, g["op_a4_grad_v"], **TOL)
    np.testing.assert_allclose(gt, g["op_a4_grad_table"], rtol=1e-4, atol=1e-4)
    # v1 form (v/3 + table per axis) is the same function
    out1 = ref.attention_step2_with_rel_pos_value(g["op_a3_out"], g["op_v"], g["blk0_index_0"], g["blk0_index_1"],
                                                  g["wa_table_v"], g["blk0_rel_idx_cpu"], n_out=g["op_v"].shape[0])
    np.testing.assert_allclose(out1, g["op_a4_out"], **TOL)
    ga1, gv1, gt1 = ref.attention_step2_with_rel_pos_value_backward(g["op_a4_grad_out"], g["op_a3_out"], g["op_v"], g["blk0_index_0"],
                                                                    g["blk0_index_1"], g["wa_table_v"], g["blk0_rel_idx_cpu"])
    np.testing.assert_allclose(ga1, g["op_a3_grad_out"], **TOL)
    np.testing.assert_allclose(gv1, g["op_a4_grad_v"], **TOL)
    np.testing.assert_allclose(gt1, g["op_a4_grad_table"], rtol=1e-4, atol=1e-4)


def test_attention_step2_plain_is_rel_value_with_zero_table(golden):
    g = golden
    zero = np.zeros_like(g["wa_table_v"])
    a = ref.attention_step2(g["op_a3_out"], g["op_v"], g["blk0_index_0"], g["blk0_index_1"], n_out=g["op_v"].shape[0])
    b = ref.attention_step2_with_rel_pos_value_v2(g["op_a3_out"], g["op_v"], g["blk0_offsets"], g["blk0_index_1"], zero, g["blk0_rel_idx_cpu"])
    np.testing.assert_allclose(a, b, **TOL)
    ga, gv = ref.attention_step2_backward(g["op_a4_grad_out"], g["op_a3_out"], g["op_v"], g["blk0_index_0"], g["blk0_index_1"])
    ga2, gv2, _ = ref.attention_step2_with_rel_pos_value_v2_backward(g["op_a4_grad_out"], g["op_a3_out"], g["op_v"], g["blk0_offsets"],
                                                                     g["blk0_index_1"], zero, g["blk0_rel_idx_cpu"])
    np.testing.assert_allclose(ga, ga2, **TOL)
    np.testing.assert_allclose(gv, gv2, **TOL)


# ---- FPS / KNN: no Python form and no test in the reference => parity unpinned; cross-checks ----
def _naive_fps(xyz, start, end, m):
    """Textbook FPS in float64 on one batch element; ties (absent in random data) aside, the
    argmax sequence must equal the kernel restatement's."""
    p = xyz[start:end].astype(np.float64)
    dist = np.full(len(p), 1e10)
    out = [0]
    for _ in range(m - 1):
        d = ((p - p[out[-1]]) ** 2).sum(1)
        dist = np.minimum(dist, d)
        out.append(int(dist.argmax()))
    return np.asarray(out) + start


def test_fps_against_naive():
    rng = np.random.default_rng(0)
    xyz = rng.random((3000, 3), dtype=np.float32)
    offset = np.array([1300, 3000], dtype=np.int32)
    new_offset = np.array([200, 420], dtype=np.int32)
    idx = ref.furthestsampling(xyz, offset, new_offset)
    assert np.array_equal(idx[:200], _naive_fps(xyz, 0, 1300, 200))
    assert np.array_equal(idx[200:], _naive_fps(xyz, 1300, 3000, 220))


def test_fps_tie_rule_tree_order_then_first_in_thread():
    """Exact ties are resolved by the reduction tree of sampling_cuda_kernel.cu:64-123: slot t absorbs
    slot t+s and keeps its own entry on ties, for s = B/2 ... 1.  Net effect: among equal distances the
    thread with the smallest BIT-REVERSED id wins, and inside one thread the first (lowest) k (:57)."""
    # B = opt_n_threads(6) = 4 -> thread t owns k = t and t+4
    xyz = np.zeros((6, 3), dtype=np.float32)
    xyz[1, 0] = 1.0          # k=1 -> thread 1
    xyz[5, 1] = 1.0          # k=5 -> thread 1 as well: first in thread wins
    idx = ref.furthestsampling(xyz, np.array([6], np.int32), np.array([2], np.int32))
    assert idx.tolist() == [0, 1]
    xyz = np.zeros((6, 3), dtype=np.float32)
    xyz[2, 0] = 1.0          # k=2 -> thread 2 (bit-reversed 1)
    xyz[5, 1] = 1.0          # k=5 -> thread 1 (bit-reversed 2): loses although its thread id is lower
    idx = ref.furthestsampling(xyz, np.array([6], np.int32), np.array([2], np.int32))
    assert idx.tolist() == [0, 2]
    xyz = np.zeros((6, 3), dtype=np.float32)
    xyz[3, 0] = 1.0          # thread 3 (bit-reversed 3)
    xyz[5, 1] = 1.0          # thread 1 (bit-reversed 2)
    idx = ref.furthestsampling(xyz, np.array([6], np.int32), np.array([2], np.int32))
    assert idx.tolist() == [0, 5]


def test_knn_against_cdist():
    rng = np.random.default_rng(1)
    xyz = rng.random((2000, 3), dtype=np.float32)
    new_xyz = rng.random((300, 3), dtype=np.float32)
    offset = np.array([900, 2000], dtype=np.int32)
    new_offset = np.array([100, 300], dtype=np.int32)
    idx, dist = ref.knnquery(16, xyz, new_xyz, offset, new_offset)
    for lo, hi, qlo, qhi in ((0, 900, 0, 100), (900, 2000, 100, 300)):
        d = torch.cdist(torch.from_numpy(new_xyz[qlo:qhi]).double(), torch.from_numpy(xyz[lo:hi]).double())
        top = d.topk(16, dim=1, largest=False)
        assert np.array_equal(idx[qlo:qhi], top.indices.numpy() + lo)
        np.testing.assert_allclose(dist[qlo:qhi], top.values.numpy(), rtol=1e-5, atol=1e-6)


def test_knn_fewer_points_than_k():
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], dtype=np.float32)
    idx, dist = ref.knnquery(5, xyz, xyz[:1], np.array([3], np.int32), np.array([1], np.int32))
    # unfilled heap slots keep (1e10, start) and sort last (knnquery_cuda_kernel.cu:88-91,103)
    assert idx[0].tolist() == [0, 1, 2, 0, 0]
    np.testing.assert_allclose(dist[0], [0, 1, 2, 1e5, 1e5])


# ---- Swin3D variant: the oracle's restatement against vectors produced by model/swin3d_transformer.py ----
def test_swin3d_oracle_index_matches_reference_golden():
    import os
    from oracle import index_ref
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "swin3d_window_attention.npz"))
    xyz = torch.from_numpy(g["xyz"])
    for pat in (0, 1):
        got = index_ref.swin_stage_indices(xyz, g["offset"], float(g["window_size"]), float(g["quant_size"]), pat)
        assert np.array_equal(got["index_0"].numpy(), g[f"p{pat}_index_0"].astype(np.int64))
        assert np.array_equal(got["index_1"].numpy(), g[f"p{pat}_index_1"].astype(np.int64))
        assert np.array_equal(got["offsets"].numpy(), g[f"p{pat}_offsets"].astype(np.int64))
        assert got["n_max"] == int(g[f"p{pat}_n_max"])
        assert np.array_equal(got["rel_idx"].numpy(), g[f"p{pat}_rel_idx_cpu"].astype(np.int32))
        assert got["rel_idx"].min() >= 0 and got["rel_idx"].max() <= 30


def test_stratified_h6_L80_oracle_matches_reference_golden():
    """second fixture of the reference's WindowAttention: 4 000 points (BASELINE config-1 size), h = 6, L = 80"""
    import os
    from oracle import index_ref, pointops_ref as ref
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "window_attention_4000_h6.npz"))
    xyz = torch.from_numpy(g["xyz"])
    w, quant = float(g["window_size"]), float(g["quant_size"])
    got = index_ref.build_stage_indices(xyz, g["offset"], w, quant, torch.from_numpy(g["downsample_idx"].astype(np.int32)), 0, div_mode="cpu")
    assert np.array_equal(got["index_0"].numpy(), g["index_0"].astype(np.int64))
    assert np.array_equal(got["index_1"].numpy(), g["index_1"].astype(np.int64))
    assert np.array_equal(got["offsets"].numpy(), g["offsets"].astype(np.int64))
    assert np.array_equal(got["rel_idx"].numpy(), g["rel_idx_cpu"].astype(np.int32))
    # the module through the oracle's kernels: qkv Linear, A1, A2, softmax, A4, proj
    N, C = g["feats"].shape
    h = g["table_q"].shape[1]
    qkv = (g["feats"] @ g["qkv_weight"].T + g["qkv_bias"]).reshape(N, 3, h, C // h).transpose(1, 0, 2, 3)
    q, k, v = (np.ascontiguousarray(qkv[i], dtype=np.float32) for i in range(3))
    q = q * np.float32((C // h) ** -0.5)
    i1, offs, rel = g["index_1"].astype(np.int32), g["offsets"].astype(np.int32), g["rel_idx_cpu"].astype(np.int32)
    sm = ref.segment_softmax(ref.attention_step1_v2(q, k, i1, offs) + ref.dot_prod_with_idx_v3(q, offs, k, i1, g["table_q"], g["table_k"], rel), offs)
    x = ref.attention_step2_with_rel_pos_value_v2(sm, v, offs, i1, g["table_v"], rel)
    y = x.reshape(N, C) @ g["proj_weight"].T + g["proj_bias"]
    np.testing.assert_allclose(y, g["out"], rtol=1e-4, atol=1e-4)


def test_voxelize_crop_and_data_prepare_match_reference():
    """SURVEY 8f-2, pinned: oracle/index_ref.py's voxelize / crop_nearest / data_prepare against tests/golden/voxelize_crop.npz,
    which make_golden_dataprep.py produced by executing the reference's util/voxelize.py and util/data_util.py (stable argsort,
    recorded random draws).  Everything bit-exact: keys, indices, counts, cropped coordinates, features, labels."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "voxelize_crop.npz"))
    for tag in ("f64", "f32"):
        coord = g[f"{tag}_coord"]
        dt = coord.dtype
        feat, label = g[f"{tag}_feat"].astype(dt), g[f"{tag}_label"].astype(dt)
        vox = dt.type(0.04)
        assert np.array_equal(index_ref.fnv_hash_vec(np.floor(coord / vox)), g[f"{tag}_keys"]), tag
        assert np.array_equal(index_ref.voxelize(coord, vox, 0, g[f"{tag}_train_rand"].astype(np.int64)), g[f"{tag}_train_idx"]), tag
        idx_sort, count = index_ref.voxelize(coord, vox, 1)
        assert np.array_equal(idx_sort, g[f"{tag}_val_idx_sort"]) and np.array_equal(count, g[f"{tag}_val_count"]), tag
        for fn, div in (("data_prepare_v101", 255.0), ("data_prepare_scannet", None)):
            for split in ("val", "train"):
                key = f"{tag}_{fn}_{split}"
                seed = int(g[key + "_seed"]) if split == "train" else None
                c, f, l = index_ref.data_prepare(coord, feat, label, split, 0.04, 4000, g[key + "_rand"].astype(np.int64), seed, div)
                assert c.dtype == np.float32 and f.dtype == np.float32 and l.dtype == np.int64
                assert np.array_equal(c, g[key + "_coord"]) and np.array_equal(f, g[key + "_feat"]) and np.array_equal(l, g[key + "_label"]), key
