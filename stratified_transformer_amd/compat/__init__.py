"""Providers of the third-party names `model/stratified_transformer.py:3-7` imports.

None of these packages is installed (or installable) on the MI355X boxes, and two of them sit on
the hot path, so the build supplies them:

  torch_scatter.scatter_softmax   -> HIP segment softmax (stratified_transformer_amd/csrc/softmax.hip)
                                     for the model's call (2-D src, ascending 1-D index, dim=0); a
                                     generic torch formulation for every other call
  torch_geometric.nn.voxel_grid   -> the torch_cluster grid_cluster arithmetic, evaluated with torch
                                     ops on the tensor's device
  timm.models.layers              -> DropPath, trunc_normal_
  torch_points3d ... KPConvLayer / FastBatchNorm1d -> import-only placeholders: the KPConv stem is
                                     outside the hot path; using them raises.

`install()` only fills names that are NOT importable, so a real installation always wins.
"""
import importlib
import sys
import types

import torch


# ---------------------------------------------------------------------------------------------
# torch_scatter
# ---------------------------------------------------------------------------------------------
def _scatter_softmax_generic(src, index, dim, eps):
    """torch_scatter 2.0.6 composite/softmax.py with torch ops (any device, any index order)."""
    dim = dim if dim >= 0 else src.dim() + dim
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape).expand_as(src)
    size = list(src.shape)
    size[dim] = int(index.max()) + 1 if index.numel() else 0
    mx = torch.full(size, float("-inf"), dtype=src.dtype, device=src.device).scatter_reduce(dim, index, src, reduce="amax", include_self=True)
    ex = (src - mx.gather(dim, index)).exp()
    sm = torch.zeros(size, dtype=src.dtype, device=src.device).scatter_add_(dim, index, ex)
    return ex / (sm + eps).gather(dim, index)


_ASSUME_MODEL_ORDER = False


def assume_model_call_order(flag=True):
    """Declare that scatter_softmax is only called the way model/stratified_transformer.py:183-205 calls it (right after
    attention_step1_v2 / dot_prod_with_idx_v3 on the same pair list, `index` = the sorted per-pair query ids): the shim then
    never synchronises the host.  Off by default: the shim then keeps torch_scatter's full semantics for any index."""
    global _ASSUME_MODEL_ORDER
    _ASSUME_MODEL_ORDER = bool(flag)


def scatter_softmax(src, index, dim=-1, eps=1e-12):
    """Same signature as torch_scatter.scatter_softmax.  The model's call site
    (model/stratified_transformer.py:205: src [M,h] fp32 on the GPU, index = ascending index_0,
    dim=0) runs on the HIP segment-softmax kernel; everything else takes the generic path."""
    if not torch.is_floating_point(src):
        raise ValueError("`scatter_softmax` can only be computed over tensors with floating point data types.")
    d = dim if dim >= 0 else src.dim() + dim
    if src.is_cuda and src.dim() == 2 and index.dim() == 1 and d == 0 and src.dtype == torch.float32 and index.numel() > 0:
        from .. import _lib, pointops as P
        # The model calls this right after A1 / A2 on the same pair list (:183-205): the CSR offsets those operators just
        # used PROBABLY describe `index` - a hint (same device, same M, tensor not written to since), never trusted: one
        # kernel checks that the offsets tile [0, M) and that every pair of segment i carries the id i (csr_matches; it
        # cannot read outside the two tensors whatever a stale hint holds).  The verdict is read back - one host sync,
        # which torch_scatter itself pays for `index.max()` - unless the caller has declared the model's call order with
        # assume_model_call_order(True): then nothing is read back and a mismatch poisons the result's first row with NaN
        # instead of taking the generic path (the segment kernels clamp every bound into [0, M], so even then nothing
        # is read or written out of bounds).  Either way the offsets are not rebuilt.
        offsets = P.last_csr(src.device.index, index.numel())
        if offsets is not None and index.dtype in (torch.int32, torch.int64):
            same = P.csr_matches(offsets, index)
            if _ASSUME_MODEL_ORDER:
                return P.segment_softmax(src, offsets, same)
            if bool(same):
                return P.segment_softmax(src, offsets)
        # ascending index <=> segments are contiguous runs; one host sync, like the model's own asserts (:189-190)
        if bool((index[1:] >= index[:-1]).all()):
            from ..pointops import segment_softmax
            _, counts = torch.unique_consecutive(index, return_counts=True)
            offsets = torch.zeros(counts.shape[0] + 1, dtype=torch.int32, device=src.device)
            offsets[1:] = counts.cumsum(0)
            return segment_softmax(src, offsets)
    return _scatter_softmax_generic(src, index, dim, eps)


# ---------------------------------------------------------------------------------------------
# torch_geometric.nn.voxel_grid (1.7.0) -> torch_cluster.grid_cluster
# ---------------------------------------------------------------------------------------------
def voxel_grid(pos, batch, size, start=None, end=None):
    """cluster id per point: voxel_d = (int64)((pos_d - start_d) / size_d) in fp32, batch appended as a
    4th coordinate of cell size 1; id = sum_d voxel_d * prod_{e<d} ((int64)((end_e-start_e)/size_e) + 1)."""
    pos = pos.unsqueeze(-1) if pos.dim() == 1 else pos
    dim = pos.shape[1]

    def rep(v):
        if v is None:
            return None
        v = v.tolist() if torch.is_tensor(v) else v
        return list(v) if isinstance(v, (list, tuple)) else [v] * dim

    size, start, end = rep(size), rep(start), rep(end)
    pos4 = torch.cat([pos, batch.unsqueeze(-1).type_as(pos)], dim=-1)
    size4 = torch.tensor(size + [1], dtype=pos.dtype, device=pos.device)
    start4 = pos4.min(0)[0] if start is None else torch.tensor(start + [0], dtype=pos.dtype, device=pos.device)
    end4 = pos4.max(0)[0] if end is None else torch.tensor(end + [int(batch.max())], dtype=pos.dtype, device=pos.device)
    vox = ((pos4 - start4) / size4).to(torch.int64)
    nvox = ((end4 - start4) / size4).to(torch.int64) + 1
    mult = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.int64, device=pos.device), nvox[:-1]]), 0)
    return (vox * mult).sum(-1)


# ---------------------------------------------------------------------------------------------
# timm.models.layers
# ---------------------------------------------------------------------------------------------
class DropPath(torch.nn.Module):
    """Stochastic depth per sample (timm 0.4.9 semantics)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if not self.drop_prob or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = (keep + torch.rand(shape, dtype=x.dtype, device=x.device)).floor_()
        return x.div(keep) * mask


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return torch.nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class _OffPath(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()

    def forward(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__}: KPConv stem is outside the hot path (SURVEY.md §8); install torch_points3d to use it")


class KPConvLayer(_OffPath):
    pass


class FastBatchNorm1d(_OffPath):
    pass


def _importable(name):
    try:
        importlib.import_module(name)
        return True
    except Exception:
        return False


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__dict__["__stratified_transformer_amd_shim__"] = True
    sys.modules[name] = m
    return m


def install():
    if not _importable("torch_scatter"):
        _module("torch_scatter", scatter_softmax=scatter_softmax)
    if not _importable("torch_geometric.nn"):
        tg = _module("torch_geometric")
        tg.nn = _module("torch_geometric.nn", voxel_grid=voxel_grid)
    if not _importable("timm.models.layers"):
        t = _module("timm")
        t.models = _module("timm.models")
        t.models.layers = _module("timm.models.layers", DropPath=DropPath, trunc_normal_=trunc_normal_)
    if not _importable("torch_points3d.modules.KPConv.kernels"):
        tp = _module("torch_points3d")
        tp.modules = _module("torch_points3d.modules")
        tp.modules.KPConv = _module("torch_points3d.modules.KPConv")
        tp.modules.KPConv.kernels = _module("torch_points3d.modules.KPConv.kernels", KPConvLayer=KPConvLayer)
        tp.core = _module("torch_points3d.core")
        tp.core.common_modules = _module("torch_points3d.core.common_modules", FastBatchNorm1d=FastBatchNorm1d)
