"""Wall time of every single pass of a run (diagnostic; GPU box only): outliers and what they coincide with."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
K = int(sys.argv[1]) if len(sys.argv) > 1 else 40
import gc
for spec in (True, False):
    states, _ = pipeline.scene_pass(xyz, off, cfg, fused="cell", speculate=spec)
    for _ in range(3):
        pipeline.scene_pass(xyz, off, cfg, states, fused="cell", speculate=spec)
    torch.cuda.synchronize()
    gc.collect(); gc.disable()
    ts, before = [], dict(pipeline.SPECULATION)
    for _ in range(K):
        t0 = time.perf_counter()
        pipeline.scene_pass(xyz, off, cfg, states, fused="cell", speculate=spec)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    gc.enable()
    print("speculate", spec, "reruns", pipeline.SPECULATION["reruns"] - before["reruns"], "ms:", " ".join("%.1f" % t for t in ts))
