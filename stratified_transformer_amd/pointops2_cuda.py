"""Stand-in for the reference's compiled extension module `pointops2_cuda`
(lib/pointops2/src/pointops_api.cpp:16-45): the same function names, positional arguments and
ownership rules (the caller allocates and zero-fills every output; functions return None), taking
torch tensors and forwarding their device pointers to the C ABI of libpointops2_hip.so.

`stratified_transformer_amd.install()` registers this module as `sys.modules["pointops2_cuda"]`, so
the reference's own `lib/pointops2/functions/pointops.py` (`import pointops2_cuda as pointops_cuda`,
pointops.py:11) binds to it unchanged.

Stricter than the reference shims (which validate nothing, e.g. attention_cuda_v2.cpp:7-16): dtype,
device and contiguity are checked, launches go to torch's current stream under a device guard, and
native errors are raised as RuntimeError.
"""
import torch

from . import _lib
from ._lib import ptr

F32, I32 = torch.float32, torch.int32


def _chk(*pairs):
    for t, dt, name in pairs:
        _lib.check_tensor(t, dt, name)


def _call(name, ref, *args):
    if ref.device.index == torch.cuda.current_device():  # the usual case: no device switch, no guard object
        _lib.call(name, *args, device=ref.device)
    else:
        with torch.cuda.device(ref.device):
            _lib.call(name, *args, device=ref.device)


def _rows(table):
    _lib.lib().pointops2_set_table_rows(int(table.shape[0]))


# sampling/sampling_cuda.cpp
def furthestsampling_cuda(b, n, xyz, offset, new_offset, tmp, idx):
    _chk((xyz, F32, "xyz"), (offset, I32, "offset"), (new_offset, I32, "new_offset"), (tmp, F32, "tmp"), (idx, I32, "idx"))
    _call("furthestsampling_cuda_launcher", xyz, int(b), int(n), ptr(xyz), ptr(offset), ptr(new_offset), ptr(tmp), ptr(idx))


# knnquery/knnquery_cuda.cpp
def knnquery_cuda(m, nsample, xyz, new_xyz, offset, new_offset, idx, dist2):
    _chk((xyz, F32, "xyz"), (new_xyz, F32, "new_xyz"), (offset, I32, "offset"), (new_offset, I32, "new_offset"),
         (idx, I32, "idx"), (dist2, F32, "dist2"))
    _call("knnquery_cuda_launcher", xyz, int(m), int(nsample), ptr(xyz), ptr(new_xyz), ptr(offset), ptr(new_offset), ptr(idx), ptr(dist2))


# grouping/grouping_cuda.cpp
def grouping_forward_cuda(m, nsample, c, input, idx, output):
    _chk((input, F32, "input"), (idx, I32, "idx"), (output, F32, "output"))
    _call("grouping_forward_cuda_launcher", input, int(m), int(nsample), int(c), ptr(input), ptr(idx), ptr(output))


def grouping_backward_cuda(m, nsample, c, grad_output, idx, grad_input):
    _chk((grad_output, F32, "grad_output"), (idx, I32, "idx"), (grad_input, F32, "grad_input"))
    _call("grouping_backward_cuda_launcher", grad_output, int(m), int(nsample), int(c), ptr(grad_output), ptr(idx), ptr(grad_input))


# interpolation/interpolation_cuda.cpp
def interpolation_forward_cuda(n, c, k, input, idx, weight, output):
    _chk((input, F32, "input"), (idx, I32, "idx"), (weight, F32, "weight"), (output, F32, "output"))
    _call("interpolation_forward_cuda_launcher", input, int(n), int(c), int(k), ptr(input), ptr(idx), ptr(weight), ptr(output))


def interpolation_backward_cuda(n, c, k, grad_output, idx, weight, grad_input):
    _chk((grad_output, F32, "grad_output"), (idx, I32, "idx"), (weight, F32, "weight"), (grad_input, F32, "grad_input"))
    _call("interpolation_backward_cuda_launcher", grad_output, int(n), int(c), int(k), ptr(grad_output), ptr(idx), ptr(weight), ptr(grad_input))


# attention/attention_cuda.cpp
def attention_step1_forward_cuda(N, M, h, C, q, k, index0, index1, attn):
    _chk((q, F32, "q"), (k, F32, "k"), (index0, I32, "index0"), (index1, I32, "index1"), (attn, F32, "attn"))
    _call("attention_step1_forward_cuda_launcher", q, int(N), int(M), int(h), int(C), ptr(q), ptr(k), ptr(index0), ptr(index1), ptr(attn))


def attention_step1_backward_cuda(N, M, h, C, grad_out, index0, index1, q, k, grad_q, grad_k):
    _chk((grad_out, F32, "grad_out"), (index0, I32, "index0"), (index1, I32, "index1"), (q, F32, "q"), (k, F32, "k"),
         (grad_q, F32, "grad_q"), (grad_k, F32, "grad_k"))
    _call("attention_step1_backward_cuda_launcher", q, int(N), int(M), int(h), int(C), ptr(grad_out), ptr(index0), ptr(index1),
          ptr(q), ptr(k), ptr(grad_q), ptr(grad_k))


def attention_step2_forward_cuda(N, M, h, C, attn, v, index0, index1, output):
    _chk((attn, F32, "attn"), (v, F32, "v"), (index0, I32, "index0"), (index1, I32, "index1"), (output, F32, "output"))
    _call("attention_step2_forward_cuda_launcher", v, int(N), int(M), int(h), int(C), ptr(attn), ptr(v), ptr(index0), ptr(index1), ptr(output))


def attention_step2_backward_cuda(N, M, h, C, grad_out, index0, index1, attn, v, grad_attn, grad_v):
    _chk((grad_out, F32, "grad_out"), (index0, I32, "index0"), (index1, I32, "index1"), (attn, F32, "attn"), (v, F32, "v"),
         (grad_attn, F32, "grad_attn"), (grad_v, F32, "grad_v"))
    _call("attention_step2_backward_cuda_launcher", v, int(N), int(M), int(h), int(C), ptr(grad_out), ptr(index0), ptr(index1),
          ptr(attn), ptr(v), ptr(grad_attn), ptr(grad_v))


# attention_v2/attention_cuda_v2.cpp
def attention_step1_forward_cuda_v2(N, M, h, C, n_max, q, k, index0_offsets, index1, attn):
    _chk((q, F32, "q"), (k, F32, "k"), (index0_offsets, I32, "index0_offsets"), (index1, I32, "index1"), (attn, F32, "attn"))
    _call("attention_step1_forward_cuda_launcher_v2", q, int(N), int(M), int(h), int(C), int(n_max), ptr(q), ptr(k),
          ptr(index0_offsets), ptr(index1), ptr(attn))


def attention_step1_backward_cuda_v2(N, M, h, C, n_max, grad_out, index0_offsets, index1, q, k, grad_q, grad_k):
    _chk((grad_out, F32, "grad_out"), (index0_offsets, I32, "index0_offsets"), (index1, I32, "index1"), (q, F32, "q"), (k, F32, "k"),
         (grad_q, F32, "grad_q"), (grad_k, F32, "grad_k"))
    _call("attention_step1_backward_cuda_launcher_v2", q, int(N), int(M), int(h), int(C), int(n_max), ptr(grad_out),
          ptr(index0_offsets), ptr(index1), ptr(q), ptr(k), ptr(grad_q), ptr(grad_k))


def attention_step2_forward_cuda_v2(N, M, h, C, attn, v, index0, index1, output):
    attention_step2_forward_cuda(N, M, h, C, attn, v, index0, index1, output)


def attention_step2_backward_cuda_v2(N, M, h, C, grad_out, index0, index1, attn, v, grad_attn, grad_v):
    attention_step2_backward_cuda(N, M, h, C, grad_out, index0, index1, attn, v, grad_attn, grad_v)


# rpe/relative_pos_encoding_cuda.cpp
def dot_prod_with_idx_forward_cuda(N, M, h, hdim, q, index, table, rel_idx, output):
    _chk((q, F32, "q"), (index, I32, "index"), (table, F32, "table"), (rel_idx, I32, "rel_idx"), (output, F32, "output"))
    _call("dot_prod_with_idx_forward_cuda_launcher", q, int(N), int(M), int(h), int(hdim), ptr(q), ptr(index), ptr(table), ptr(rel_idx), ptr(output))


def dot_prod_with_idx_backward_cuda(N, M, h, hdim, grad_out, q, index, table, rel_idx, grad_q, grad_table):
    _chk((grad_out, F32, "grad_out"), (q, F32, "q"), (index, I32, "index"), (table, F32, "table"), (rel_idx, I32, "rel_idx"),
         (grad_q, F32, "grad_q"), (grad_table, F32, "grad_table"))
    _call("dot_prod_with_idx_backward_cuda_launcher", q, int(N), int(M), int(h), int(hdim), ptr(grad_out), ptr(q), ptr(index),
          ptr(table), ptr(rel_idx), ptr(grad_q), ptr(grad_table))


def attention_step2_with_rel_pos_value_forward_cuda(N, M, h, hdim, attn, v, index0, index1, table, rel_idx, output):
    _chk((attn, F32, "attn"), (v, F32, "v"), (index0, I32, "index0"), (index1, I32, "index1"), (table, F32, "table"),
         (rel_idx, I32, "rel_idx"), (output, F32, "output"))
    _call("attention_step2_with_rel_pos_value_forward_cuda_launcher", v, int(N), int(M), int(h), int(hdim), ptr(attn), ptr(v),
          ptr(index0), ptr(index1), ptr(table), ptr(rel_idx), ptr(output))


def attention_step2_with_rel_pos_value_backward_cuda(N, M, h, hdim, grad_out, index0, index1, attn, v, table, rel_idx,
                                                     grad_attn, grad_v, grad_table):
    _chk((grad_out, F32, "grad_out"), (index0, I32, "index0"), (index1, I32, "index1"), (attn, F32, "attn"), (v, F32, "v"),
         (table, F32, "table"), (rel_idx, I32, "rel_idx"), (grad_attn, F32, "grad_attn"), (grad_v, F32, "grad_v"), (grad_table, F32, "grad_table"))
    _call("attention_step2_with_rel_pos_value_backward_cuda_launcher", v, int(N), int(M), int(h), int(hdim), ptr(grad_out),
          ptr(index0), ptr(index1), ptr(attn), ptr(v), ptr(table), ptr(rel_idx), ptr(grad_attn), ptr(grad_v), ptr(grad_table))


# rpe_v2/relative_pos_encoding_cuda_v2.cpp
def dot_prod_with_idx_forward_cuda_v2(N, M, h, hdim, n_max, T, q, index_q, k, index_k, table_q, table_k, rel_idx,
                                      rel_idx_offsets, sort_indices, output):
    _chk((q, F32, "q"), (index_q, I32, "index_q"), (k, F32, "k"), (index_k, I32, "index_k"), (table_q, F32, "table_q"),
         (table_k, F32, "table_k"), (rel_idx, I32, "rel_idx"), (output, F32, "output"))
    _call("dot_prod_with_idx_forward_cuda_launcher_v2", q, int(N), int(M), int(h), int(hdim), int(n_max), int(T), ptr(q), ptr(index_q),
          ptr(k), ptr(index_k), ptr(table_q), ptr(table_k), ptr(rel_idx), ptr(rel_idx_offsets), ptr(sort_indices), ptr(output))


def dot_prod_with_idx_backward_cuda_v2(N, M, h, hdim, n_max, T, grad_out, q, index_q, k, index_k, table_q, table_k, rel_idx,
                                       rel_idx_offsets, sort_indices, grad_q, grad_k, grad_table_q, grad_table_k):
    _chk((grad_out, F32, "grad_out"), (q, F32, "q"), (index_q, I32, "index_q"), (k, F32, "k"), (index_k, I32, "index_k"),
         (table_q, F32, "table_q"), (table_k, F32, "table_k"), (rel_idx, I32, "rel_idx"), (grad_q, F32, "grad_q"),
         (grad_k, F32, "grad_k"), (grad_table_q, F32, "grad_table_q"), (grad_table_k, F32, "grad_table_k"))
    _call("dot_prod_with_idx_backward_cuda_launcher_v2", q, int(N), int(M), int(h), int(hdim), int(n_max), int(T), ptr(grad_out), ptr(q),
          ptr(index_q), ptr(k), ptr(index_k), ptr(table_q), ptr(table_k), ptr(rel_idx), ptr(rel_idx_offsets), ptr(sort_indices),
          ptr(grad_q), ptr(grad_k), ptr(grad_table_q), ptr(grad_table_k))


def dot_prod_with_idx_forward_cuda_v3(N, M, h, hdim, n_max, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx, output):
    _chk((q, F32, "q"), (index_q_offsets, I32, "index_q_offsets"), (k, F32, "k"), (index_k, I32, "index_k"),
         (table_q, F32, "table_q"), (table_k, F32, "table_k"), (rel_idx, I32, "rel_idx"), (output, F32, "output"))
    _rows(table_q)
    _call("dot_prod_with_idx_forward_cuda_launcher_v3", q, int(N), int(M), int(h), int(hdim), int(n_max), ptr(q), ptr(index_q_offsets),
          ptr(k), ptr(index_k), ptr(table_q), ptr(table_k), ptr(rel_idx), ptr(output))


def dot_prod_with_idx_backward_cuda_v3(N, M, h, hdim, n_max, grad_out, q, index_q_offsets, k, index_k, table_q, table_k, rel_idx,
                                       grad_q, grad_k, grad_table_q, grad_table_k):
    _chk((grad_out, F32, "grad_out"), (q, F32, "q"), (index_q_offsets, I32, "index_q_offsets"), (k, F32, "k"), (index_k, I32, "index_k"),
         (table_q, F32, "table_q"), (table_k, F32, "table_k"), (rel_idx, I32, "rel_idx"), (grad_q, F32, "grad_q"),
         (grad_k, F32, "grad_k"), (grad_table_q, F32, "grad_table_q"), (grad_table_k, F32, "grad_table_k"))
    _rows(table_q)
    _call("dot_prod_with_idx_backward_cuda_launcher_v3", q, int(N), int(M), int(h), int(hdim), int(n_max), ptr(grad_out), ptr(q),
          ptr(index_q_offsets), ptr(k), ptr(index_k), ptr(table_q), ptr(table_k), ptr(rel_idx), ptr(grad_q), ptr(grad_k),
          ptr(grad_table_q), ptr(grad_table_k))


def attention_step2_with_rel_pos_value_forward_cuda_v2(N, M, h, hdim, n_max, attn, v, index0_offsets, index1, table, rel_idx, output):
    _chk((attn, F32, "attn"), (v, F32, "v"), (index0_offsets, I32, "index0_offsets"), (index1, I32, "index1"), (table, F32, "table"),
         (rel_idx, I32, "rel_idx"), (output, F32, "output"))
    _rows(table)
    _call("attention_step2_with_rel_pos_value_forward_cuda_launcher_v2", v, int(N), int(M), int(h), int(hdim), int(n_max), ptr(attn), ptr(v),
          ptr(index0_offsets), ptr(index1), ptr(table), ptr(rel_idx), ptr(output))


def attention_step2_with_rel_pos_value_backward_cuda_v2(N, M, h, hdim, n_max, grad_out, index0_offsets, index1, attn, v, table, rel_idx,
                                                        grad_attn, grad_v, grad_table):
    _chk((grad_out, F32, "grad_out"), (index0_offsets, I32, "index0_offsets"), (index1, I32, "index1"), (attn, F32, "attn"),
         (v, F32, "v"), (table, F32, "table"), (rel_idx, I32, "rel_idx"), (grad_attn, F32, "grad_attn"), (grad_v, F32, "grad_v"),
         (grad_table, F32, "grad_table"))
    _rows(table)
    _call("attention_step2_with_rel_pos_value_backward_cuda_launcher_v2", v, int(N), int(M), int(h), int(hdim), int(n_max), ptr(grad_out),
          ptr(index0_offsets), ptr(index1), ptr(attn), ptr(v), ptr(table), ptr(rel_idx), ptr(grad_attn), ptr(grad_v), ptr(grad_table))


# subtraction/ and aggregation/ (Point-Transformer vector attention) are bound by the reference
# (pointops_api.cpp:23-26) but called by no model in it: not on the hot path.
def _off_path(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name}: Point-Transformer op outside the Stratified hot path (SURVEY.md §8)")
    f.__name__ = name
    return f


subtraction_forward_cuda = _off_path("subtraction_forward_cuda")
subtraction_backward_cuda = _off_path("subtraction_backward_cuda")
aggregation_forward_cuda = _off_path("aggregation_forward_cuda")
aggregation_backward_cuda = _off_path("aggregation_backward_cuda")
