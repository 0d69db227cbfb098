"""bench.py's single_pass leg, with variations of the timer (diagnostic; GPU box only)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline

cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
SEED = int(os.environ.get("LEG_SEED", "0"))
states, _ = pipeline.scene_pass(xyz, off, cfg, None, None, seed=SEED, fused="cell")


KEEP = []


def leg(fused, only, steps=20, warm=5):
    st = states
    if os.environ.get("LEG_EMPTY"):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    for _ in range(warm):
        st, res = pipeline.scene_pass(xyz, off, cfg, st, fused=fused)
    live = pipeline.Timer(True, only=only) if only is not None else None
    torch.cuda.synchronize()
    ms0 = torch.cuda.memory_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        st, res = pipeline.scene_pass(xyz, off, cfg, st, live, fused=fused)
    torch.cuda.synchronize()
    keep = os.environ.get("LEG_KEEP", "")
    if "live" in keep:
        KEEP.append(live)
    if "res" in keep:
        KEEP.append(res)
    ms = (time.perf_counter() - t0) / steps * 1e3
    b = sum(e0.elapsed_time(e1) for n, e0, e1, _ in live.spans if n.startswith("attn_bwd")) / steps if live else float('nan')
    ms1 = torch.cuda.memory_stats()
    print('fused=%-6s only=%-26s pass %.2f ms  attn_bwd %.2f ms | device allocs %d frees %d, reserved %.2f GB allocated %.2f GB' % (
        fused, only, ms, b, ms1["num_device_alloc"] - ms0["num_device_alloc"], ms1["num_device_free"] - ms0["num_device_free"],
        ms1["reserved_bytes.all.current"] / 2**30, ms1["allocated_bytes.all.current"] / 2**30), flush=True)


ORDER = os.environ.get("LEG_ORDER", "ops,cell,cell,cell").split(",")
for name in ORDER:
    leg({"ops": False, "cell": "cell", "fwd": "cell_fwd", "bf16": "cell_bf16"}[name], ("attn", "fps/", "comm/"))
