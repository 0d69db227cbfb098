"""Where and when the workgroups of the stage-0 forward cell kernel run, alone and beside 16 held CUs (needs the CA_TRACE build of
the library: P2_LIB_PATH=.../libp2_trace.so; diagnostic, GPU box only)."""
import os, sys, time, ctypes
import numpy as np
import torch
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline, fused, _lib

N = 100000
cfg = pipeline.s3dis_config()
xyz = torch.from_numpy(scene.make_room(N, 0)).cuda()
off = torch.tensor([N], dtype=torch.int32, device='cuda')
states, results = pipeline.scene_pass(xyz, off, cfg, cells=True)
torch.cuda.synchronize()
s, r = states[0], results[0]
tq, tk, tv = s.tables
plan = r['even'].cells
L = _lib.lib()
L.pointops2_diag_hold_cus_launcher.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
word = torch.zeros(64 + 64 * 1024, dtype=torch.int32, device='cuda')
side = torch.cuda.Stream()
buf = np.zeros(3 * 1024, dtype=np.uint64)


def run(label, hold):
    for it in range(3):
        torch.cuda.synchronize()
        if hold:
            with torch.cuda.stream(side):
                L.pointops2_set_stream(ctypes.c_void_p(side.cuda_stream))
                L.pointops2_diag_hold_cus_launcher(16, 4000, 0, ctypes.c_void_p(word.data_ptr()), hold)
            time.sleep(0.0015)
        with torch.no_grad():
            fused.cell_attention(s.q, s.k, s.v, tq, tk, tv, plan)
        torch.cuda.synchronize()
    L.pointops2_diag_read_cell_trace(buf.ctypes.data_as(ctypes.c_void_p))
    t = buf.reshape(-1, 3)[:255]
    t0, t1, hw = t[:, 0].astype(np.int64), t[:, 1].astype(np.int64), t[:, 2]
    base = t0.min()
    st, en = (t0 - base) / 100.0, (t1 - base) / 100.0  # us (100 MHz)
    place = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw & np.uint64(0xffffffff)) >> np.uint64(8) & np.uint64(0xff))
    print('%-22s kernel %.0f us | starts: median %.0f max %.0f us, started later than 20 us: %d | duration median %.0f max %.0f | distinct places %d'
          % (label, en.max(), np.median(st), st.max(), int((st > 20).sum()), np.median(en - st), (en - st).max(), len(set(place.tolist()))))
    late = np.argsort(st)[-5:]
    print('    latest starts (us):', [round(float(st[i])) for i in late], 'their durations:', [round(float(en[i] - st[i])) for i in late])


run('alone', 0)
run('16 CUs held (120K LDS)', 120)
run('16 WGs beside (8K LDS)', 8)
