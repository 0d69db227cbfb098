"""How much do the cell attention kernels of stage 0 lose when the round sampler (16 CUs) or the kNN query runs beside them?
(diagnostic; GPU box only)"""
import os, sys, threading, time
import torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline, fused, pointops as P

N = 100000
cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(N, 0)).cuda()
off = torch.tensor([N], dtype=torch.int32, device='cuda')
states, results = pipeline.scene_pass(xyz, off, cfg, cells=True)
torch.cuda.synchronize()
s, r = states[0], results[0]
tq, tk, tv = s.tables
plan = r['even'].cells
n4 = torch.tensor([N // 4 + 1], dtype=torch.int32, device='cuda')
sub = xyz[P.furthestsampling(xyz, off, n4).long()].contiguous()
side = torch.cuda.Stream()


def measure(label, beside=None):
    """`beside` enqueues a few ms of work on the side stream (no host thread: the host only launches)"""
    tf = tb = 0.0
    reps = 10
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    for it in range(reps + 2):
        torch.cuda.synchronize()
        if beside is not None:
            with torch.cuda.stream(side):
                beside()
            time.sleep(0.0015)  # (the sampler's set-up kernels and head are through; its rounds are running)
        e[0].record()
        out = fused.cell_attention(s.q, s.k, s.v, tq, tk, tv, plan)
        e[1].record()
        out.backward(s.grad_out)
        e[2].record()
        torch.cuda.synchronize()
        if it >= 2:
            tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    print('%-28s fwd us %5.0f bwd us %5.0f' % (label, tf / reps * 1e3, tb / reps * 1e3), flush=True)


def fps():
    P.clear_caches()
    P.furthestsampling(xyz, off, n4)


import ctypes
from stratified_transformer_amd import _lib
L = _lib.lib()
L.pointops2_diag_hold_cus_launcher.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
L.pointops2_diag_hold_cus_launcher.restype = None
word = torch.zeros(64 + 64 * 1024, dtype=torch.int32, device='cuda')


def hold(blocks, mode, lds_kb):
    def f():
        L.pointops2_set_stream(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.pointops2_diag_hold_cus_launcher(blocks, 4000, mode, ctypes.c_void_p(word.data_ptr()), lds_kb)
    return f


measure('alone')
measure('beside the round sampler', fps)
measure('16 CUs held (spin, 120K LDS)', hold(16, 0, 120))
measure('16 WGs beside (spin, 8K LDS)', hold(16, 0, 8))
measure('16 WGs polling a word', hold(16, 1, 8))
measure('16 WGs sc1 store/load', hold(16, 2, 8))
measure('alone again')
