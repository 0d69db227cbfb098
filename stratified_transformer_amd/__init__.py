"""stratified_transformer_amd — MI355X (gfx950) implementation of the Stratified Transformer's
windowed sparse-attention hot path, behind the reference's own operator API.

    stratified_transformer_amd.pointops        drop-in for lib/pointops2/functions/pointops.py
    stratified_transformer_amd.pointops2_cuda  drop-in for the compiled module `pointops2_cuda`
    stratified_transformer_amd.compat          providers of the third-party names the model imports
                                               (torch_scatter.scatter_softmax, torch_geometric.nn.voxel_grid, ...)
    stratified_transformer_amd.layers          installable fast BasicLayer.forward / WindowAttention.forward (same signatures)
    include/pointops2_hip.h                    the C ABI underneath (libpointops2_hip.so)

`install()` registers the drop-in modules in sys.modules so that the reference's
model/stratified_transformer.py imports and runs unmodified under PyTorch-ROCm.
"""
import sys

__all__ = ["install", "build"]


def build(verbose=False):
    """Compile libpointops2_hip.so for gfx950 (works without a GPU)."""
    from . import _lib
    return _lib.build(verbose=verbose)


def install(third_party=True, fast_layers=False):
    """Make `import pointops2_cuda`, `from lib.pointops2.functions import pointops` and (optionally) the
    model's third-party imports resolve to this package.

    fast_layers=True: additionally rebind `BasicLayer.forward` / `WindowAttention.forward` of the (unmodified, importable)
    `model.stratified_transformer` to the forms of `stratified_transformer_amd.layers`: the stage's index is built once on
    the device and every attention block runs as one fused function on its cell plan - the path bench.py's headline
    (`single_pass.cell`) measures.  Without it the model runs on the operator API alone (`single_pass.operator_api`)."""
    from . import pointops2_cuda
    sys.modules.setdefault("pointops2_cuda", pointops2_cuda)
    if third_party:
        from . import compat
        compat.install()
    if fast_layers:
        from . import layers
        return layers.install_fast_layers()
