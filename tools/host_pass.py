"""Host-side cost of enqueuing one scene pass (no timer events), and a cProfile of it (diagnostic; GPU box only)."""
import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
FUSED = 'cell' if os.environ.get('SPAN_CELL', '1') == '1' else False
states, _ = pipeline.scene_pass(xyz, off, cfg, fused=FUSED)
for _ in range(2):
    pipeline.scene_pass(xyz, off, cfg, states, fused=FUSED)
torch.cuda.synchronize()
for it in range(3):
    t0 = time.perf_counter()
    pipeline.scene_pass(xyz, off, cfg, states, fused=FUSED)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host enqueue %.2f ms, pass %.2f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
pipeline.scene_pass(xyz, off, cfg, states, fused=FUSED)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
