"""Host time to ENQUEUE one attention block (forward + backward through the op API) vs its device time
(diagnostic; GPU box only)."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
states, results = pipeline.scene_pass(xyz, off, cfg)
torch.cuda.synchronize()
timer = pipeline.Timer(False)
for si in (0, 2, 3):
    s, r = states[si], results[si]
    for _ in range(3):
        pipeline.attention_block(s, r['even'], timer)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pipeline.attention_block(s, r['even'], timer)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('stage %d: host enqueue %.3f ms/block, wall %.3f ms/block' % (si, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
s, r = states[3], results[3]
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    pipeline.attention_block(s, r['even'], timer)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
