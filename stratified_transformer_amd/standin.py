"""TEST / BENCH SCAFFOLDING (not the product): containers with the attribute names of the reference's model classes (model/stratified_transformer.py:
Mlp :66, TransitionDown :87, WindowAttention :114, SwinTransformerBlock :219, BasicLayer :250), so that the installable layer
forwards of stratified_transformer_amd.layers can be exercised - and timed, bench.py `installed_layers` - on the GPU box, where the
reference itself cannot travel.
`BasicLayer.forward` / `WindowAttention.forward` are deliberately ABSENT here (they raise): the tests install the package's
own forms with layers.patch_classes().  The block and the transition keep the call structure their reference counterparts have
(pre-norm residual block; FPS -> kNN grouping -> norm / linear / max-pool), because that is the caller the installed forwards
must work under; state-dict keys equal the reference's, so golden parameters load by name.
"""
import torch
from torch import nn


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1, self.act, self.fc2 = nn.Linear(dim, hidden), nn.GELU(), nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, quant_size):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.quant_size = dim, num_heads, window_size, quant_size
        self.scale = (dim // num_heads) ** -0.5
        self.rel_query = self.rel_key = self.rel_value = True
        self.quant_grid_length = int((2 * window_size + 1e-4) // quant_size)
        shape = (2 * self.quant_grid_length, num_heads, dim // num_heads, 3)
        self.relative_pos_query_table = nn.Parameter(torch.zeros(shape))
        self.relative_pos_key_table = nn.Parameter(torch.zeros(shape))
        self.relative_pos_value_table = nn.Parameter(torch.zeros(shape))
        self.qkv, self.proj, self.proj_drop = nn.Linear(dim, 3 * dim), nn.Linear(dim, dim), nn.Dropout(0.0)

    def forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max):
        raise NotImplementedError("stand-in: install stratified_transformer_amd.layers.window_attention_forward")


class SwinTransformerBlock(nn.Module):
    def __init__(self, dim, num_heads, window_size, quant_size, mlp_ratio=4.0):
        super().__init__()
        self.norm1, self.norm2 = nn.LayerNorm(dim), nn.LayerNorm(dim)
        self.attn = WindowAttention(dim, window_size, num_heads, quant_size)
        self.drop_path = nn.Identity()
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, feats, xyz, index_0, index_1, index_0_offsets, n_max):
        feats = feats + self.drop_path(self.attn(self.norm1(feats), xyz, index_0, index_1, index_0_offsets, n_max))
        return feats + self.drop_path(self.mlp(self.norm2(feats)))


class TransitionDown(nn.Module):
    def __init__(self, in_channels, out_channels, ratio, k):
        super().__init__()
        self.ratio, self.k = ratio, k
        self.norm, self.linear, self.pool = nn.LayerNorm(in_channels), nn.Linear(in_channels, out_channels, bias=False), nn.MaxPool1d(k)

    def forward(self, feats, xyz, offset):
        from stratified_transformer_amd import index_build, pointops
        n_offset = torch.tensor(index_build.transition_down_offset(offset.tolist(), self.ratio), dtype=torch.int32, device=xyz.device)
        idx = pointops.furthestsampling(xyz, offset, n_offset)
        n_xyz = xyz[idx.long(), :]
        g = pointops.queryandgroup(self.k, xyz, n_xyz, feats, None, offset, n_offset, use_xyz=False)
        m, k, c = g.shape
        g = self.linear(self.norm(g.view(m * k, c)).view(m, k, c)).transpose(1, 2).contiguous()
        return self.pool(g).squeeze(-1), n_xyz, n_offset


class BasicLayer(nn.Module):
    def __init__(self, downsample_scale, depth, channel, num_heads, window_size, quant_size, ratio=0.25, k=16, out_channels=None):
        super().__init__()
        self.depth, self.window_size, self.downsample_scale = depth, window_size, downsample_scale
        self.blocks = nn.ModuleList([SwinTransformerBlock(channel, num_heads, window_size, quant_size) for _ in range(depth)])
        self.downsample = TransitionDown(channel, out_channels, ratio, k) if out_channels else None

    def forward(self, feats, xyz, offset):
        raise NotImplementedError("stand-in: install stratified_transformer_amd.layers.basic_layer_forward")
