"""ctypes binding of libpointops2_hip.so (C ABI: include/pointops2_hip.h).

The product path has no CPU fallback: if the HIP library is missing, `lib()` raises.  PyTorch is used
only for device memory and streams; every pointer crossing this boundary is a raw device address.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("P2_LIB_PATH") or os.path.join(_HERE, "lib", "libpointops2_hip.so")  # (P2_LIB_PATH: kernel experiments)
CSRC = os.path.join(_HERE, "csrc")
_lib = None

I, U, P, Z, F = ctypes.c_int, ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_float

# name -> argtypes; mirrors include/pointops2_hip.h one to one
SIGNATURES = {
    "pointops2_set_stream": [P],
    "pointops2_diag_set_fps_patience": [ctypes.c_ulonglong],
    "pointops2_set_table_rows": [I],
    "pointops2_set_workspace": [P, Z],
    "pointops2_set_point_count": [I],
    "pointops2_set_batch_count": [I],
    "pointops2_set_key_rows": [I],
    "pointops2_set_fps_resume": [P, P],
    "pointops2_set_fps_hint": [I],
    "pointops2_set_csc": [P, P, P],
    "pointops2_csc_build": [I, I, P, P, P, P, P, P, Z],
    "furthestsampling_cuda_launcher": [I, I, P, P, P, P, P],
    "knnquery_cuda_launcher": [I, I, P, P, P, P, P, P],
    "grouping_forward_cuda_launcher": [I, I, I, P, P, P],
    "grouping_backward_cuda_launcher": [I, I, I, P, P, P],
    "interpolation_forward_cuda_launcher": [I, I, I, P, P, P, P],
    "interpolation_backward_cuda_launcher": [I, I, I, P, P, P, P],
    "attention_step1_forward_cuda_launcher": [I, I, I, I, P, P, P, P, P],
    "attention_step1_backward_cuda_launcher": [I, I, I, I, P, P, P, P, P, P, P],
    "attention_step2_forward_cuda_launcher": [I, I, I, I, P, P, P, P, P],
    "attention_step2_backward_cuda_launcher": [I, I, I, I, P, P, P, P, P, P, P],
    "attention_step1_forward_cuda_launcher_v2": [I, I, I, I, U, P, P, P, P, P],
    "attention_step1_backward_cuda_launcher_v2": [I, I, I, I, U, P, P, P, P, P, P, P],
    "attention_step2_forward_cuda_launcher_v2": [I, I, I, I, P, P, P, P, P],
    "attention_step2_backward_cuda_launcher_v2": [I, I, I, I, P, P, P, P, P, P, P],
    "dot_prod_with_idx_forward_cuda_launcher": [I, I, I, I, P, P, P, P, P],
    "dot_prod_with_idx_backward_cuda_launcher": [I, I, I, I, P, P, P, P, P, P, P],
    "attention_step2_with_rel_pos_value_forward_cuda_launcher": [I, I, I, I, P, P, P, P, P, P, P],
    "attention_step2_with_rel_pos_value_backward_cuda_launcher": [I, I, I, I, P, P, P, P, P, P, P, P, P, P],
    "dot_prod_with_idx_forward_cuda_launcher_v2": [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P],
    "dot_prod_with_idx_backward_cuda_launcher_v2": [I, I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P],
    "dot_prod_with_idx_forward_cuda_launcher_v3": [I, I, I, I, I, P, P, P, P, P, P, P, P],
    "dot_prod_with_idx_backward_cuda_launcher_v3": [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P, P, P],
    "attention_step2_with_rel_pos_value_forward_cuda_launcher_v2": [I, I, I, I, I, P, P, P, P, P, P, P],
    "attention_step2_with_rel_pos_value_backward_cuda_launcher_v2": [I, I, I, I, I, P, P, P, P, P, P, P, P, P, P],
    "subtraction_forward_cuda_launcher": [I, I, I, P, P, P, P],
    "subtraction_backward_cuda_launcher": [I, I, I, P, P, P, P],
    "aggregation_forward_cuda_launcher": [I, I, I, I, P, P, P, P, P],
    "aggregation_backward_cuda_launcher": [I, I, I, I, P, P, P, P, P, P, P, P],
    "segment_softmax_forward_launcher": [I, I, I, P, P, P],
    "window_logits_softmax_forward_launcher": [I, I, I, I, P, P, P, P, P, P, P, P],
    "window_attention_backward_launcher": [I, I, I, I] + [P] * 18,
    "segment_softmax_backward_launcher": [I, I, I, P, P, P, P],
    "csr_expand_launcher": [I, I, P, P],
    "pointops2_csr_matches_launcher": [I, I, P, P, I, P],
    "pointops2_bbox_launcher": [I, P, P],
    "pointops2_window_partition_launcher": [I, I, P, P, P, F, F, I, P, P, P, P, P, Z],
    "pointops2_window_partitions4_launcher": [I, I, P, P, P, F, P, P, P, P, P, P, Z],
    "pointops2_row_order_launcher": [I, I, P, P, P, P, Z],
    "pointops2_set_row_order": [P, I],
    "pointops2_window_coord_launcher": [I, P, P, F, I, P],
    "pointops2_sampled_buckets_launcher": [I, I, P, P, P, P, P, P, P, P, Z],
    "pointops2_pairs_count_launcher": [I, P, P, P, P, P, P, P, P, Z],
    "pointops2_pairs_fill_launcher": [I, P, F, F, P, P, P, P, P, P, P, P, P, P, P],
    "pointops2_voxel_keys_launcher": [I, I, P, ctypes.c_double, P],
    "pointops2_crop_dist_launcher": [I, I, P, I, P],
    "pointops2_cell_plan_count_launcher": [I, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, Z],
    "pointops2_cell_plan_prepare_launcher": [I, I, P, P, P, P, P, P, Z],
    "pointops2_cell_plan_sizes_launcher": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, Z],
    "pointops2_cell_plan_fill_launcher": [I, P, F, F, I, P, P, P, P, P, P, P, P, P, P, P, P],
    "cell_attention_forward_launcher": [P, I, I, I] + [P] * 9,
    "cell_attention_backward_launcher": [P, I, I, I] + [P] * 16,
    "cell_attention_forward_bf16_launcher": [P, I, I, I] + [P] * 9,
    "cell_attention_backward_bf16_launcher": [P, I, I, I] + [P] * 16,
}
# entry points with a non-void result
RESULTS = {
    "pointops2_get_stream": ([], P),
    "pointops2_last_error": ([], ctypes.c_char_p),
    "pointops2_abi_version": ([], I),
    "pointops2_csc_workspace_bytes": ([I, I], Z),
    "pointops2_fps_workspace_bytes": ([I, I], Z),
    "pointops2_knn_workspace_bytes": ([I, I, I], Z),
    "pointops2_index_workspace_bytes": ([I], Z),
    "pointops2_partitions4_workspace_bytes": ([I], Z),
    "pointops2_row_order_workspace_bytes": ([I], Z),
    "pointops2_cell_plan_workspace_bytes": ([I], Z),
}


def exported_symbols():
    """Every symbol include/pointops2_hip.h declares (used by the no-GPU ABI test)."""
    return sorted(list(SIGNATURES) + list(RESULTS))


def build(verbose=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j8"] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C stratified_transformer_amd/csrc`). "
                "There is no CPU fallback for the product path.")
        l = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = None
        for name, (argtypes, restype) in RESULTS.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = restype
        _lib = l
    return _lib


def ptr(t):
    """Raw device address of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def check_tensor(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the pointops2 HIP path has no CPU fallback), got {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


CALLS = [0]  # library launcher calls so far (bench.py: per pass)


def call(name, *args, device=None):
    """Launch `name` on torch's current stream of `device` and surface library errors."""
    CALLS[0] += 1
    l = lib()
    if _raw_stream is not None and device is not None and device.index is not None:
        stream = _raw_stream(device.index)  # the raw hipStream_t, without building a torch.cuda.Stream object
    else:
        stream = torch.cuda.current_stream(device).cuda_stream
    l.pointops2_set_stream(stream)
    getattr(l, name)(*args)
    err = l.pointops2_last_error()
    if err is not None:
        raise RuntimeError(f"{name}: {err.decode()}")
