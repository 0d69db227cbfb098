"""Un-profiled timeline of one scene pass from the pipeline's own event spans (diagnostic; GPU box only)."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")  # as bench.py
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
FUSED = 'cell' if os.environ.get('SPAN_CELL', '1') == '1' else False
states, _ = pipeline.scene_pass(xyz, off, cfg, fused=FUSED)
pipeline.scene_pass(xyz, off, cfg, states, fused=FUSED)
torch.cuda.synchronize()
timer = pipeline.Timer(True)
ref = torch.cuda.Event(enable_timing=True)
ref.record()
import time
h0 = time.perf_counter()
pipeline.scene_pass(xyz, off, cfg, states, timer, fused=FUSED)
h1 = time.perf_counter()
timer.run("mark/tool_after_pass", lambda: None)
torch.cuda.current_stream().synchronize()
h1b = time.perf_counter()
timer.run("mark/tool_after_stream_sync", lambda: None)
end = torch.cuda.Event(enable_timing=True)
end.record()
torch.cuda.synchronize()
print("host: pass returned at %.2f ms, main stream drained at %.2f ms" % ((h1 - h0) * 1e3, (h1b - h0) * 1e3))
h2 = time.perf_counter()
print('pass %.2f ms by events (host enqueue %.2f ms, host wall incl. the final synchronize %.2f ms)' % (ref.elapsed_time(end), (h1 - h0) * 1e3, (h2 - h0) * 1e3))
rows = [(ref.elapsed_time(e0), ref.elapsed_time(e1), name) for name, e0, e1, _ in timer.spans]
rows.sort()
cur = None
for s, e, n in rows:
    key = n.split('/')[0] if n.startswith('attn') else n
    if cur and cur[2] == key and s - cur[1] < 0.3:
        cur[1] = max(cur[1], e); cur[3] += 1
    else:
        if cur: print('%7.2f -> %7.2f  %-16s x%d' % tuple(cur))
        cur = [s, e, key, 1]
print('%7.2f -> %7.2f  %-16s x%d' % tuple(cur))
