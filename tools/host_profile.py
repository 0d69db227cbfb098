"""Where the HOST spends a scene pass (cProfile over K passes; diagnostic, GPU box only).

    python tools/host_profile.py [single|inflight] [K]
"""
import cProfile, os, pstats, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

mode = sys.argv[1] if len(sys.argv) > 1 else "single"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device="cuda")
import gc
gc.disable()
if mode == "single":
    states, _ = pipeline.scene_pass(xyz, off, cfg, fused="cell")
    for _ in range(2):
        pipeline.scene_pass(xyz, off, cfg, states, fused="cell")
    torch.cuda.synchronize()
    def work():
        for _ in range(K):
            pipeline.scene_pass(xyz, off, cfg, states, fused="cell")
        torch.cuda.synchronize()
else:
    lanes = []
    for li in range(4):
        ls = torch.cuda.Stream()
        ls.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(ls):
            st, _ = pipeline.scene_pass(xyz, off, cfg, None, None, seed=1234, lane=li, fused="cell")
        lanes.append((ls, st))
    pipeline.passes_in_flight([xyz], [off], cfg, lanes, 2, fused="cell", offset_host_list=[[100000]])
    torch.cuda.synchronize()
    def work():
        pipeline.passes_in_flight([xyz], [off], cfg, lanes, K, fused="cell", offset_host_list=[[100000]])
        torch.cuda.synchronize()
t0 = time.perf_counter(); work(); t1 = time.perf_counter()
print("%s: %.2f ms per pass un-profiled" % (mode, (t1 - t0) / K * 1e3))
pr = cProfile.Profile()
pr.enable(); work(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumulative").print_stats(45)
