"""Host time of the geometry phase (scene_pass_phases, first next()) per call site (diagnostic; GPU box only)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pointops as P, pipeline

log = []
def wrap(name, fn):
    def f(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); log.append((name, (time.perf_counter() - t) * 1e3)); return r
    return f
P.furthestsampling = wrap("fps", P.furthestsampling)
P.knnquery = wrap("knn", P.knnquery)
cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device="cuda")
st, _ = pipeline.scene_pass(xyz, off, cfg)
pipeline.scene_pass(xyz, off, cfg, st)
torch.cuda.synchronize()
for rep in range(2):
    log.clear()
    t = time.perf_counter()
    g = pipeline.scene_pass_phases(xyz, off, cfg, st)
    next(g)
    t1 = time.perf_counter()
    print("phase 1 host ms", round((t1 - t) * 1e3, 2), [(n, round(v, 2)) for n, v in log])
    try:
        next(g)
    except StopIteration:
        pass
    print("phase 2 host ms", round((time.perf_counter() - t1) * 1e3, 2))
    torch.cuda.synchronize()
