import torch, sys
sys.path.insert(0, '.')
from stratified_transformer_amd import scene, pointops as P
xyz = torch.from_numpy(scene.make_room(100000, 0)).cuda()
off = torch.tensor([100000], dtype=torch.int32, device='cuda')
P.clear_caches()
idx = P.furthestsampling(xyz, off, torch.tensor([12501], dtype=torch.int32, device='cuda'))
torch.cuda.synchronize()
