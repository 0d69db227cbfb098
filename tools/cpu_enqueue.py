"""How long does the host need to ENQUEUE one scene pass, compared with the pass itself? (diagnostic; GPU box only)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
states, _ = pipeline.scene_pass(xyz, off, cfg)
torch.cuda.synchronize()
for it in range(5):
    t0 = time.perf_counter()
    pipeline.scene_pass(xyz, off, cfg, states)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('enqueue %.1f ms   pass %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for it in range(3):
    pipeline.scene_pass(xyz, off, cfg, states)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
