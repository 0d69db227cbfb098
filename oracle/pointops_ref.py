"""TEST INFRASTRUCTURE — ctypes front-end of liboracle.so (see pointops_oracle.c).

All functions take and return CPU numpy arrays (float32 / int32, C-contiguous) and follow the
calling convention of the reference's Python wrappers (lib/pointops2/functions/pointops.py):
the wrapper allocates and zero-fills outputs, the native side only fills them.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pointops_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().oracle_num_threads())


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def host_cores(cap=16):
    """CPU share of this process (affinity mask), capped: a one-GPU box owns 16 of the host's cores."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, cap))


def opt_n_threads(n):
    return int(lib().oracle_opt_n_threads(int(n)))


def furthestsampling(xyz, offset, new_offset):
    """pointops.py:14-29"""
    xyz, offset, new_offset = _f(xyz), _i(offset), _i(new_offset)
    n, b = xyz.shape[0], offset.shape[0]
    n_max = int(offset[0])
    for i in range(1, b):
        n_max = max(int(offset[i] - offset[i - 1]), n_max)
    idx = np.zeros(int(new_offset[b - 1]), dtype=np.int32)
    tmp = np.full(n, 1e10, dtype=np.float32)
    lib().oracle_furthestsampling(b, n_max, _p(xyz), _p(offset), _p(new_offset), _p(tmp), _p(idx))
    return idx


def knnquery(nsample, xyz, new_xyz, offset, new_offset):
    """pointops.py:34-47 (returns sqrt of the kernel's squared distances, like the wrapper)"""
    xyz = _f(xyz)
    new_xyz = xyz if new_xyz is None else _f(new_xyz)
    offset, new_offset = _i(offset), _i(new_offset)
    m = new_xyz.shape[0]
    idx = np.zeros((m, nsample), dtype=np.int32)
    dist2 = np.zeros((m, nsample), dtype=np.float32)
    lib().oracle_knnquery(m, nsample, _p(xyz), _p(new_xyz), _p(offset), _p(new_offset), _p(idx), _p(dist2))
    return idx, np.sqrt(dist2)


def ball_query(radius, max_num, x, y, offset_x, offset_y):
    """train_backup.py:362-364 `tp.ball_query(radius, max_num, x, y, mode="partial_dense", ...)[0]` (torch_points_kernels 0.6.10:
    third-party, absent from the reference tree - PARITY UNPINNED; restated from its published behaviour): per point of y up
    to max_num points of x of the same batch element with squared distance < radius^2, nearest first, padded with -1.
    Brute force: every candidate's squared distance with the kNN kernel's fma chain, a stable sort by (distance, index)."""
    x, y = _f(x), _f(y)
    offset_x, offset_y = _i(offset_x), _i(offset_y)
    r2 = np.float32(radius) * np.float32(radius)
    m = y.shape[0]
    idx = np.full((m, max_num), -1, np.int32)
    d2o = np.full((m, max_num), -1.0, np.float32)
    xs, ys = 0, 0
    for b in range(len(offset_x)):
        xe, ye = int(offset_x[b]), int(offset_y[b])
        xb = x[xs:xe].astype(np.float64)
        for i in range(ys, ye):
            d = (y[i].astype(np.float64) - xb)  # exact differences in f64, then the f32 fma chain: fma(dz,dz, fma(dx,dx, dy*dy))
            dx, dy, dz = (d[:, a].astype(np.float32) for a in range(3))
            t = (dy * dy).astype(np.float32)
            t = (dx.astype(np.float64) * dx.astype(np.float64) + t.astype(np.float64)).astype(np.float32)
            dd = (dz.astype(np.float64) * dz.astype(np.float64) + t.astype(np.float64)).astype(np.float32)
            order = np.lexsort((np.arange(xe - xs), dd))[:max_num]
            keep = order[dd[order] < r2]
            idx[i, : len(keep)] = keep + xs
            d2o[i, : len(keep)] = dd[keep]
        xs, ys = xe, ye
    return idx, d2o


def attention_step1_v2(q, k, index1, index0_offsets):
    q, k, index1, index0_offsets = _f(q), _f(k), _i(index1), _i(index0_offsets)
    N, h, d = q.shape
    M = index1.shape[0]
    out = np.zeros((M, h), dtype=np.float32)
    lib().oracle_attention_step1_forward_v2(index0_offsets.shape[0] - 1, M, h, h * d, _p(q), _p(k), _p(index0_offsets), _p(index1), _p(out))
    return out


def attention_step1_v2_backward(grad_out, q, k, index1, index0_offsets):
    grad_out, q, k, index1, index0_offsets = _f(grad_out), _f(q), _f(k), _i(index1), _i(index0_offsets)
    N, h, d = q.shape
    M = index1.shape[0]
    gq = np.zeros_like(q)
    gk = np.zeros_like(k)
    lib().oracle_attention_step1_backward_v2(N, M, h, h * d, _p(grad_out), _p(index0_offsets), _p(index1), _p(q), _p(k), _p(gq), _p(gk))
    return gq, gk


def attention_step1(q, k, index0, index1):
    q, k, index0, index1 = _f(q), _f(k), _i(index0), _i(index1)
    N, h, d = q.shape
    M = index0.shape[0]
    out = np.zeros((M, h), dtype=np.float32)
    lib().oracle_attention_step1_forward(k.shape[0], M, h, h * d, _p(q), _p(k), _p(index0), _p(index1), _p(out))
    return out


def attention_step1_backward(grad_out, q, k, index0, index1):
    grad_out, q, k, index0, index1 = _f(grad_out), _f(q), _f(k), _i(index0), _i(index1)
    N, h, d = q.shape
    M = index0.shape[0]
    gq, gk = np.zeros_like(q), np.zeros_like(k)
    lib().oracle_attention_step1_backward(N, M, h, h * d, _p(grad_out), _p(index0), _p(index1), _p(q), _p(k), _p(gq), _p(gk))
    return gq, gk


def attention_step2(attn, v, index0, index1, n_out=None):
    attn, v, index0, index1 = _f(attn), _f(v), _i(index0), _i(index1)
    M, h = attn.shape
    d = v.shape[2]
    n_out = int(index0.max()) + 1 if n_out is None else n_out
    out = np.zeros((n_out, h, d), dtype=np.float32)
    lib().oracle_attention_step2_forward(n_out, M, h, h * d, _p(attn), _p(v), _p(index0), _p(index1), _p(out))
    return out


def attention_step2_backward(grad_out, attn, v, index0, index1):
    grad_out, attn, v, index0, index1 = _f(grad_out), _f(attn), _f(v), _i(index0), _i(index1)
    M, h = attn.shape
    d = v.shape[2]
    ga, gv = np.zeros_like(attn), np.zeros_like(v)
    lib().oracle_attention_step2_backward(grad_out.shape[0], M, h, h * d, _p(grad_out), _p(index0), _p(index1), _p(attn), _p(v), _p(ga), _p(gv))
    return ga, gv


def dot_prod_with_idx_v3(q, index_q_offsets, k, index_k, table_q, table_k, rel_idx):
    q, k, table_q, table_k = _f(q), _f(k), _f(table_q), _f(table_k)
    index_q_offsets, index_k, rel_idx = _i(index_q_offsets), _i(index_k), _i(rel_idx)
    N, h, d = q.shape
    M = index_k.shape[0]
    out = np.zeros((M, h), dtype=np.float32)
    lib().oracle_dot_prod_with_idx_forward_v3(N, M, h, d, _p(q), _p(index_q_offsets), _p(k), _p(index_k), _p(table_q), _p(table_k), _p(rel_idx), _p(out))
    return out


def dot_prod_with_idx_v3_backward(grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx):
    grad_out, q, k, table_q, table_k = _f(grad_out), _f(q), _f(k), _f(table_q), _f(table_k)
    index_q_offsets, index_k, rel_idx = _i(index_q_offsets), _i(index_k), _i(rel_idx)
    N, h, d = q.shape
    M = index_k.shape[0]
    L = table_q.shape[0]
    gq, gk, gtq, gtk = np.zeros_like(q), np.zeros_like(k), np.zeros_like(table_q), np.zeros_like(table_k)
    lib().oracle_dot_prod_with_idx_backward_v3(N, M, h, d, L, _p(grad_out), _p(q), _p(index_q_offsets), _p(k), _p(index_k),
                                               _p(table_q), _p(table_k), _p(rel_idx), _p(gq), _p(gk), _p(gtq), _p(gtk))
    return gq, gk, gtq, gtk


def dot_prod_with_idx(q, index, table, rel_idx):
    q, table, index, rel_idx = _f(q), _f(table), _i(index), _i(rel_idx)
    N, h, d = q.shape
    M = index.shape[0]
    out = np.zeros((M, h), dtype=np.float32)
    lib().oracle_dot_prod_with_idx_forward(N, M, h, d, _p(q), _p(index), _p(table), _p(rel_idx), _p(out))
    return out


def dot_prod_with_idx_backward(grad_out, q, index, table, rel_idx):
    grad_out, q, table, index, rel_idx = _f(grad_out), _f(q), _f(table), _i(index), _i(rel_idx)
    N, h, d = q.shape
    M = index.shape[0]
    gq, gt = np.zeros_like(q), np.zeros_like(table)
    lib().oracle_dot_prod_with_idx_backward(N, M, h, d, _p(grad_out), _p(q), _p(index), _p(table), _p(rel_idx), _p(gq), _p(gt))
    return gq, gt


def attention_step2_with_rel_pos_value_v2(attn, v, index0_offsets, index1, table, rel_idx):
    attn, v, table = _f(attn), _f(v), _f(table)
    index0_offsets, index1, rel_idx = _i(index0_offsets), _i(index1), _i(rel_idx)
    M, h = attn.shape
    d = v.shape[2]
    N = index0_offsets.shape[0] - 1  # CSR rows (queries); v may have more rows (keys)
    out = np.zeros((N, h, d), dtype=np.float32)
    lib().oracle_attention_step2_with_rel_pos_value_forward_v2(N, M, h, d, _p(attn), _p(v), _p(index0_offsets), _p(index1), _p(table), _p(rel_idx), _p(out))
    return out


def attention_step2_with_rel_pos_value_v2_backward(grad_out, attn, v, index0_offsets, index1, table, rel_idx):
    grad_out, attn, v, table = _f(grad_out), _f(attn), _f(v), _f(table)
    index0_offsets, index1, rel_idx = _i(index0_offsets), _i(index1), _i(rel_idx)
    M, h = attn.shape
    d = v.shape[2]
    N = index0_offsets.shape[0] - 1
    L = table.shape[0]
    ga, gv, gt = np.zeros_like(attn), np.zeros_like(v), np.zeros_like(table)
    lib().oracle_attention_step2_with_rel_pos_value_backward_v2(N, M, h, d, L, _p(grad_out), _p(index0_offsets), _p(index1), _p(attn), _p(v),
                                                                _p(table), _p(rel_idx), _p(ga), _p(gv), _p(gt))
    return ga, gv, gt


def attention_step2_with_rel_pos_value(attn, v, index0, index1, table, rel_idx, n_out=None):
    attn, v, table = _f(attn), _f(v), _f(table)
    index0, index1, rel_idx = _i(index0), _i(index1), _i(rel_idx)
    M, h = attn.shape
    d = v.shape[2]
    n_out = int(index0.max()) + 1 if n_out is None else n_out
    out = np.zeros((n_out, h, d), dtype=np.float32)
    lib().oracle_attention_step2_with_rel_pos_value_forward(n_out, M, h, d, _p(attn), _p(v), _p(index0), _p(index1), _p(table), _p(rel_idx), _p(out))
    return out


def attention_step2_with_rel_pos_value_backward(grad_out, attn, v, index0, index1, table, rel_idx):
    grad_out, attn, v, table = _f(grad_out), _f(attn), _f(v), _f(table)
    index0, index1, rel_idx = _i(index0), _i(index1), _i(rel_idx)
    M, h = attn.shape
    d = v.shape[2]
    ga, gv, gt = np.zeros_like(attn), np.zeros_like(v), np.zeros_like(table)
    lib().oracle_attention_step2_with_rel_pos_value_backward(grad_out.shape[0], M, h, d, _p(grad_out), _p(index0), _p(index1), _p(attn), _p(v),
                                                             _p(table), _p(rel_idx), _p(ga), _p(gv), _p(gt))
    return ga, gv, gt


def grouping(inp, idx):
    inp, idx = _f(inp), _i(idx)
    m, ns = idx.shape
    c = inp.shape[1]
    out = np.zeros((m, ns, c), dtype=np.float32)
    lib().oracle_grouping_forward(m, ns, c, _p(inp), _p(idx), _p(out))
    return out


def grouping_backward(grad_out, idx, n):
    grad_out, idx = _f(grad_out), _i(idx)
    m, ns, c = grad_out.shape
    gi = np.zeros((n, c), dtype=np.float32)
    lib().oracle_grouping_backward(m, ns, c, _p(grad_out), _p(idx), _p(gi))
    return gi


def interpolation_forward(inp, idx, weight):
    inp, idx, weight = _f(inp), _i(idx), _f(weight)
    n, k = idx.shape
    c = inp.shape[1]
    out = np.zeros((n, c), dtype=np.float32)
    lib().oracle_interpolation_forward(n, c, k, _p(inp), _p(idx), _p(weight), _p(out))
    return out


def interpolation_backward(grad_out, idx, weight, m):
    grad_out, idx, weight = _f(grad_out), _i(idx), _f(weight)
    n, k = idx.shape
    c = grad_out.shape[1]
    gi = np.zeros((m, c), dtype=np.float32)
    lib().oracle_interpolation_backward(n, c, k, _p(grad_out), _p(idx), _p(weight), _p(gi))
    return gi


def segment_softmax(src, offsets):
    src, offsets = _f(src), _i(offsets)
    N = offsets.shape[0] - 1
    out = np.zeros_like(src)
    lib().oracle_segment_softmax_forward(N, src.shape[1], _p(src), _p(offsets), _p(out))
    return out


def segment_softmax_backward(y, grad_y, offsets):
    y, grad_y, offsets = _f(y), _f(grad_y), _i(offsets)
    N = offsets.shape[0] - 1
    gx = np.zeros_like(y)
    lib().oracle_segment_softmax_backward(N, y.shape[1], _p(y), _p(grad_y), _p(offsets), _p(gx))
    return gx
