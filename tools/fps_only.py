"""FPS alone on the bench scene: stage-0 stratified + transition calls (diagnostic; GPU box only).  P2_FPS_STAMPS=1 prints phase cycles."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pointops as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
xyz = torch.from_numpy(scene.make_room(N, 0)).cuda()
off = torch.tensor([N], dtype=torch.int32, device='cuda')
n8 = torch.tensor([N // 8 + 1], dtype=torch.int32, device='cuda')
n4 = torch.tensor([int(N * 0.25) + 1], dtype=torch.int32, device='cuda')
for it in range(3):
    P.clear_caches()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = P.furthestsampling(xyz, off, n8)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    b = P.furthestsampling(xyz, off, n4)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print('N', N, 'stratified ms', round((t1 - t0) * 1e3, 2), 'transition (resumed) ms', round((t2 - t1) * 1e3, 2), flush=True)
