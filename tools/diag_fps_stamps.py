import torch, numpy as np, sys, time
sys.path.insert(0, '.')
from stratified_transformer_amd import scene, pointops as P
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
for m in (12501, 25001):
    for rep in range(3):
        P.clear_caches()
        torch.cuda.synchronize(); t0 = time.time()
        idx = P.furthestsampling(xyz, off, torch.tensor([m], dtype=torch.int32, device='cuda'))
        torch.cuda.synchronize()
    print('m', m, 'iters', m - 1, 'ms', (time.time() - t0) * 1e3)
