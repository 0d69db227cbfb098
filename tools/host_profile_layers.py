"""Where the HOST spends a forward + backward of the installed layers on the stand-in containers (cProfile; diagnostic, GPU box only)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline, layers, standin
from stratified_transformer_amd import pointops as P

cfg = pipeline.s3dis_config()
st = cfg.stages
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
offset = torch.tensor([100000], dtype=torch.int32, device="cuda")
net = torch.nn.ModuleList([standin.BasicLayer(cfg.downsample_scale, s.depth, s.channels, s.num_heads, s.window_size, s.quant_size, ratio=cfg.ratio, k=cfg.k,
                                              out_channels=st[i + 1].channels if i + 1 < len(st) else None) for i, s in enumerate(st)]).cuda()
feats0 = torch.randn(xyz.shape[0], st[0].channels, device="cuda")
layers.patch_classes(standin.BasicLayer, standin.WindowAttention, standin.TransitionDown)
layers.CHAIN = os.environ.get("CHAIN", "1") == "1"


def step():
    P.clear_caches(); layers.forget_clouds(); net.zero_grad(set_to_none=True)
    f, x, o = feats0.clone().requires_grad_(True), xyz, offset
    loss = None
    for layer in net:
        f_out, _, _, f, x, o = layer(f, x, o)
        term = f_out.square().mean()
        loss = term if loss is None else loss + term
    t1 = time.perf_counter()
    loss.backward()
    return t1


import gc
gc.disable()
for _ in range(3): step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter(); fw = 0.0
for _ in range(K):
    a = time.perf_counter(); b = step(); fw += b - a
torch.cuda.synchronize()
print("step %.2f ms (host: forward enqueue %.2f ms)" % ((time.perf_counter() - t0) / K * 1e3, fw / K * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(K): step()
torch.cuda.synchronize(); pr.disable()
s = pstats.Stats(pr); s.sort_stats("tottime").print_stats(30)
