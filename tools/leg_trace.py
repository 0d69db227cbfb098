"""Per-pass wall time and attention spans of back-to-back passes (bench.py's single_pass legs), diagnostic; GPU box only."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
states, _ = pipeline.scene_pass(xyz, off, cfg, fused="cell")
order = sys.argv[1].split(',') if len(sys.argv) > 1 else ["ops", "cell"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for name in order:
    fused = {"ops": False, "cell": "cell", "bf16": "cell_bf16", "fwd": "cell_fwd"}[name]
    timers, marks = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        tm = pipeline.Timer(True, only=("attn",))
        pipeline.scene_pass(xyz, off, cfg, states, tm, fused=fused)
        timers.append(tm)
        marks.append(time.perf_counter())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    per = []
    for tm in timers:
        f = sum(e0.elapsed_time(e1) for nme, e0, e1, _ in tm.spans if nme.startswith("attn_fwd"))
        b = sum(e0.elapsed_time(e1) for nme, e0, e1, _ in tm.spans if nme.startswith("attn_bwd"))
        per.append((round(f, 2), round(b, 2)))
    print(name, 'avg pass ms %.2f' % ((t1 - t0) / n * 1e3), 'host marks (ms):', [round((m - t0) * 1e3, 1) for m in marks[:6]], '...')
    print('   (fwd, bwd) ms per pass:', per)
