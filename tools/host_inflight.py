"""Host timeline of passes_in_flight: when each phase starts/ends on the host (diagnostic; GPU box only)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pointops as P, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device="cuda")
lanes = []
NL = int(sys.argv[2]) if len(sys.argv) > 2 else 3
FUSED = "cell" if os.environ.get("SPAN_CELL", "1") == "1" else False
for li in range(NL):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        st, _ = pipeline.scene_pass(xyz, off, cfg, lane=li, fused=FUSED)
    lanes.append((s, st))
torch.cuda.synchronize()
pipeline.passes_in_flight([xyz], [off], cfg, lanes, NL, fused=FUSED, offset_host_list=[[100000]])
torch.cuda.synchronize()
marks = []
calls = []
def wrap(name, fn):
    def f(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); calls.append((name, t, time.perf_counter())); return r
    return f
P.furthestsampling = wrap("fps", P.furthestsampling)
P.knnquery = wrap("knn", P.knnquery)
P.clear_caches = wrap("clear", P.clear_caches)
pipeline._offsets_tensor = wrap("upload", pipeline._offsets_tensor)
torch.cuda.Stream.wait_stream = wrap("wait_stream", torch.cuda.Stream.wait_stream)
torch.cuda.Stream.wait_event = wrap("wait_event", torch.cuda.Stream.wait_event)
torch.cuda.Event.record = wrap("ev_record", torch.cuda.Event.record)
torch.Tensor.record_stream = wrap("rec_stream", torch.Tensor.record_stream)
torch.Tensor.contiguous = wrap("contiguous", torch.Tensor.contiguous)
torch.Tensor.long = wrap("long", torch.Tensor.long)
import traceback
_tolist = torch.Tensor.tolist
def tolist_traced(self):
    t = time.perf_counter(); r = _tolist(self); d = time.perf_counter() - t
    if d > 1e-3 and len(marks) > 3:
        print("slow tolist %.1f ms at\n%s" % (d * 1e3, "".join(traceback.format_stack(limit=6)[:-1])))
    return r
torch.Tensor.tolist = tolist_traced
orig = pipeline.scene_pass_phases
def traced(*a, **k):
    g = orig(*a, **k)
    lane = k.get("lane", 0)
    def gen():
        import cProfile, pstats, io
        pr = cProfile.Profile(); pr.enable()
        t = time.perf_counter(); v = next(g); t_end = time.perf_counter()
        pr.disable()
        marks.append(("phase1 lane%d" % lane, t, t_end))
        if t_end - t > 8e-3:
            buf = io.StringIO(); pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(8)
            print("host profile of a slow phase 1:\n" + "\n".join(l for l in buf.getvalue().splitlines() if l.strip())[:3000])
        after = yield v
        t = time.perf_counter()
        try:
            g.send(after)
        except StopIteration as d:
            marks.append(("phase2 lane%d" % lane, t, time.perf_counter()))
            return d.value
    return gen()
pipeline.scene_pass_phases = traced
# finer: time the first statements of a pass
t0 = time.perf_counter()
pipeline.passes_in_flight([xyz], [off], cfg, lanes, int(sys.argv[1]) if len(sys.argv) > 1 else 6, fused=FUSED, offset_host_list=[[100000]])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
for n, a, b in marks:
    print("%-14s host %7.2f -> %7.2f ms" % (n, (a - t0) * 1e3, (b - t0) * 1e3))
for n, a, b in calls:
    if (b - a) > 0.3e-3:
        print("   slow call %-7s host %7.2f -> %7.2f ms" % (n, (a - t0) * 1e3, (b - t0) * 1e3))
print("enqueue done %.2f, device done %.2f" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
