"""The pass's kNN queries alone (diagnostic; GPU box only): k=16 of the n/4 sampled points among all, k=3 of all among the sampled."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pointops as P
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
xyz = torch.from_numpy(scene.make_room(N, 0)).cuda()
off = torch.tensor([N], dtype=torch.int32, device='cuda')
n4 = torch.tensor([N // 4 + 1], dtype=torch.int32, device='cuda')
idx = P.furthestsampling(xyz, off, n4).long()
sub = xyz[idx].contiguous()


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


print('N', N, 'k16 (%d queries in %d points) ms %.3f' % (sub.shape[0], N, timed(lambda: P.knnquery(16, xyz, sub, off, n4))),
      'k3 (%d queries in %d points) ms %.3f' % (N, sub.shape[0], timed(lambda: P.knnquery(3, sub, xyz, n4, off))))
