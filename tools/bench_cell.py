"""Cell attention only (fused.cell_attention forward + backward), per stage and pattern, on the bench scene.
Meant to be run under `rocprofv3 --kernel-trace --stats` for per-kernel times (diagnostic; GPU box only).

    python tools/bench_cell.py [N] [stages e.g. 0,1] [reps]
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stratified_transformer_amd import scene, pipeline, fused


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    stages = [int(s) for s in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 1, 2, 3]
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    cfg = pipeline.s3dis_config()
    xyz_np = scene.make_room(N, 0)
    if os.environ.get('SORTED'):  # experiment: points in (large) window order - what memory locality of the rows would give
        import numpy as np
        c = np.floor(xyz_np / float(os.environ['SORTED'])).astype(np.int64)
        xyz_np = np.ascontiguousarray(xyz_np[np.argsort((c[:, 2] * 4096 + c[:, 1]) * 4096 + c[:, 0], kind='stable')])
    xyz = torch.from_numpy(xyz_np).cuda()
    off = torch.tensor([N], dtype=torch.int32, device='cuda')
    states, results = pipeline.scene_pass(xyz, off, cfg, cells=True)
    torch.cuda.synchronize()
    for si in stages:
        s, r = states[si], results[si]
        tq, tk, tv = s.tables
        for pat in ('even', 'odd'):
            plan = r[pat].cells
            if os.environ.get('TASKS') == 'natural':  # experiment: cells in their (small window, large window) order instead of by size
                plan = plan.with_tasks(torch.arange(plan.n_cells, dtype=torch.int32, device='cuda'), torch.tensor([plan.n_cells], dtype=torch.int32, device='cuda'))
            for x in (s.q, s.k, s.v, tq, tk, tv):
                x.grad = None
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            tf = tb = 0.0
            for it in range(reps + 2):
                e[0].record()
                out = fused.cell_attention(s.q, s.k, s.v, tq, tk, tv, plan)
                e[1].record()
                out.backward(s.grad_out)
                e[2].record()
                torch.cuda.synchronize()
                if it >= 2:
                    tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
            print('stage', si, pat, 'cells', plan.n_cells, 'P', plan.n_pairs, 'K', plan.n_keyslots, 'nk_max', plan.nk_max,
                  'fwd us', round(tf / reps * 1e3), 'bwd us', round(tb / reps * 1e3), flush=True)


main()
